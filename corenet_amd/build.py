"""Builds libcorenet_hip.so (gfx950) in-tree with hipcc.  No torch involved.

  python -m corenet_amd.build        # or __graft_entry__.build()
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcorenet_hip.so")
SOURCES = ["conv_igemm.hip", "conv_inst_fwd_a.hip", "conv_inst_fwd_b.hip", "conv_inst_fwd_c.hip",
           "conv_inst_wg_a.hip", "conv_inst_wg_b.hip", "conv_inst_wg_c.hip", "batch_renorm.hip", "ray_sample.hip", "misc_ops.hip",
           "losses.hip", "fill_voxels.hip", "voxelize.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-ffp-contract=off", "-Wno-unused-result"]


def _stale(obj, src):
  if not os.path.exists(obj):
    return True
  m = os.path.getmtime(obj)
  deps = [src, os.path.join(CSRC, "crn_common.h"), os.path.join(CSRC, "conv_kernels.h"),
          os.path.join(HERE, "..", "include", "corenet_hip.h")]
  return any(os.path.getmtime(d) > m for d in deps)


def build(verbose=True, force=False):
  os.makedirs(LIBDIR, exist_ok=True)
  hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
  objs, jobs = [], []
  for s in SOURCES:
    src = os.path.join(CSRC, s)
    obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
    objs.append(obj)
    if force or _stale(obj, src):
      jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])
  def run(cmd):
    if verbose:
      print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
  with ThreadPoolExecutor(max_workers=8) as ex:
    list(ex.map(run, jobs))
  if jobs or not os.path.exists(LIB):
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
  return LIB


if __name__ == "__main__":
  build(force="--force" in sys.argv)
  print(LIB)
