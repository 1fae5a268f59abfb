#!/usr/bin/env python
"""Fails (exit 1) when a tracked profile of the current round is OLDER than the last commit that touched the kernels
(corenet_amd/csrc/): round 4 shipped a profile that contradicted the code it was supposed to document.  Run after
tools/refresh_profiles.sh and the commit of its output; hardware probes that do not depend on the library's kernels are exempt.
usage: check_profiles_fresh.py [round prefix, default r06]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix = sys.argv[1] if len(sys.argv) > 1 else "r06"
EXEMPT = ("lds_atomic_probe", "grid_barrier_probe",           # properties of the part, not of csrc/
          "defer_wgrad_experiment", "wgrad_handover",         # host-side schedule experiments (corenet_amd/model/engine.py), dated in the file
          "ray_sweep")                                        # the development sweep of the scatter (variants that no longer exist, dated in the file);
                                                              # the final kernel's numbers are r05_ray_kernels.txt / r05_small_kernels.txt
def ct(*paths):
  out = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%ct", "--"] + list(paths), capture_output=True, text=True).stdout.strip()
  return int(out) if out else 0
kernels = ct("corenet_amd/csrc")
# profiles of ONE kernel family are compared with that family's source, everything else with all of csrc/
FAMILY = (("_ray_", "corenet_amd/csrc/ray_sample.hip"), ("_mfma_neighbour", "corenet_amd/csrc/ray_sample.hip"),
          ("_voxelize", "corenet_amd/csrc/voxelize.hip"))
stale = []
for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
  if not f.startswith(prefix + "_") or any(e in f for e in EXEMPT):
    continue
  src = next((path for key, path in FAMILY if key in f), None)
  if ct(os.path.join("profiles", f)) < (ct(src) if src else kernels):
    stale.append(f)
print(f"last csrc commit {kernels}; {len(stale)} stale profile(s) of {prefix}: {stale}")
sys.exit(1 if stale else 0)
