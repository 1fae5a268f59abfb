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
SOURCES = ["conv_igemm.hip", "conv_bf3.hip", "conv_e2d.hip", "batch_renorm.hip", "ray_sample.hip", "misc_ops.hip", "losses.hip", "fill_voxels.hip",
           "voxelize.hip", "comm_rccl.hip", "stem_conv.hip", "convt_par.hip", "fill_voxels_cpu.cpp"]
# conv engine tile configurations (conv_kernels.h CRN_FWD_CONFIGS / CRN_WG_CONFIGS): one object each
CONV_CONFIGS = [(8, 1), (4, 2), (4, 1), (2, 4), (2, 2), (2, 1), (1, 4), (1, 2), (1, 1)]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-ffp-contract=off", "-Wno-unused-result"]


HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread"]


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
    obj = os.path.join(LIBDIR, os.path.splitext(s)[0] + ".o")
    objs.append(obj)
    if force or _stale(obj, src):
      flags = FLAGS if s.endswith(".hip") else HOST_FLAGS           # .cpp: host-only code of the library
      jobs.append([hipcc] + flags + ["-c", src, "-o", obj])
  inst = os.path.join(CSRC, "conv_inst.hip")
  for kind, macro in (("wgrad", "CRN_INST_WGRAD"), ("fwd", "CRN_INST_FWD")):   # the slow ones first
    for m, n in CONV_CONFIGS:
      obj = os.path.join(LIBDIR, "conv_inst_%s_%d_%d.o" % (kind, m, n))
      objs.append(obj)
      if force or _stale(obj, inst):
        jobs.append([hipcc] + FLAGS + ["-D" + macro, "-DCRN_M=%d" % m, "-DCRN_N=%d" % n, "-c", inst, "-o", obj])
  def run(cmd):
    if verbose:
      print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
  with ThreadPoolExecutor(max_workers=8) as ex:
    list(ex.map(run, jobs))
  if jobs or not os.path.exists(LIB):
    # the inline-assembly MFMA blocks of conv_bf3.hip hide their destination registers from the compiler's hazard recognizer: the
    # wait states behind them are checked on the disassembly (tools/check_mfma_hazards.py)
    chk = os.path.join(HERE, "..", "tools", "check_mfma_hazards.py")
    if os.path.exists(chk):
      run([sys.executable, chk, os.path.join(LIBDIR, "conv_bf3.o")])
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
  return LIB


TOOLS_DIR = os.path.join(HERE, "..", "tools", "_build")
TOOLS_LIB = os.path.join(TOOLS_DIR, "libcorenet_hip_tools.so")
PROBE_LIB = os.path.join(TOOLS_DIR, "libcrn_probe.so")
TOOLS_SOURCES = ["conv_igemm.hip", "conv_bf3.hip", "conv_e2d.hip", "ray_sample.hip"]     # the sources with CRN_TOOLS sections (stamp read-backs)


def build_tools(verbose=True, force=False):
  """What tools/ and the neighbour tests need beyond the product library, kept OUT of it: the MFMA hardware probe
  (tools/mfma_probe.hip -> tools/_build/libcrn_probe.so) and a second link of the library whose stamp read-backs are compiled in
  (-DCRN_TOOLS -> tools/_build/libcorenet_hip_tools.so; every other object is shared with the product build)."""
  build(verbose=verbose)
  os.makedirs(TOOLS_DIR, exist_ok=True)
  hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
  jobs, objs = [], []
  probe_src = os.path.join(HERE, "..", "tools", "mfma_probe.hip")
  if force or not os.path.exists(PROBE_LIB) or os.path.getmtime(probe_src) > os.path.getmtime(PROBE_LIB):
    jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", probe_src, "-o", PROBE_LIB])
  for s in SOURCES:
    base = os.path.splitext(s)[0]
    if s in TOOLS_SOURCES:
      obj = os.path.join(TOOLS_DIR, base + ".o")
      if force or _stale(obj, os.path.join(CSRC, s)):
        jobs.append([hipcc] + FLAGS + ["-DCRN_TOOLS", "-c", os.path.join(CSRC, s), "-o", obj])
    else:
      obj = os.path.join(LIBDIR, base + ".o")
    objs.append(obj)
  for kind in ("wgrad", "fwd"):
    for m, n in CONV_CONFIGS:
      objs.append(os.path.join(LIBDIR, "conv_inst_%s_%d_%d.o" % (kind, m, n)))
  def run(cmd):
    if verbose:
      print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
  with ThreadPoolExecutor(max_workers=8) as ex:
    list(ex.map(run, jobs))
  if jobs or not os.path.exists(TOOLS_LIB) or os.path.getmtime(LIB) > os.path.getmtime(TOOLS_LIB):
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", TOOLS_LIB] + objs + ["-ldl"])
  return TOOLS_LIB


if __name__ == "__main__":
  build(force="--force" in sys.argv)
  print(LIB)
  if "--tools" in sys.argv:
    print(build_tools(force="--force" in sys.argv))
