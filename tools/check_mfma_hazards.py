#!/usr/bin/env python
"""Build-time check of the MFMA -> VALU / VMEM / LDS hazards the compiler cannot see.

csrc/conv_bf3.hip issues the three products of a split-bf16 accumulator as ONE inline-asm block (mfma3): the compiler's hazard
recognizer does not look inside inline assembly, so it no longer knows that the block's destination registers were written by an
8-pass XDL instruction and inserts no wait states behind it.  gfx90a / gfx94x / gfx95x need software wait states between an XDL
write of a VGPR and a following VALU / VMEM / LDS / export access of it (or an MFMA that reads it as SrcA / SrcB, or overlaps it
partially as SrcC / vDst).  The numbers are LLVM's (GCNHazardRecognizer, gfx940 family), P = passes of the writing MFMA (8 for
v_mfma_f32_16x16x32_bf16): P + 3 before a VALU / VMEM / LDS / export access or an MFMA that reads the registers as SrcA / SrcB, P + 1
before an MFMA that reads them as SrcC with a different or partially overlapping vDst, 0 for the MFMA that continues the accumulation
(SrcC == vDst == the same registers: the hardware interlocks that one); the fp32 MFMAs (not XDL) need P + 2 / P.  The shipped
objects are fine because barriers and address arithmetic happen to sit in between -- this script makes that a checked property:
it disassembles the gfx950 code object inside a host object, walks the fall-through path behind every MFMA and reports every
access to its destination registers inside the window (every instruction counts one wait state, s_nop N counts N + 1; an MFMA
that continues the accumulation -- SrcC == vDst, same registers -- is the one access the hardware interlocks).
usage: check_mfma_hazards.py <object.o> [...]      exit status 1 if a hazard was found"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
PASSES = {"4x4x": 2, "16x16x4": 8, "16x16x8": 4, "16x16x16": 4, "16x16x32": 8, "16x16x64": 8, "16x16x128": 8,
          "32x32x1": 16, "32x32x2": 16, "32x32x4": 8, "32x32x8": 8, "32x32x16": 16, "32x32x64": 16}


def is_xdl(mn):
  return not re.search(r"_f32_\d+x\d+x\d+_?f32$|_f64_", mn)      # fp32 / fp64 MFMAs run on the SGEMM / DGEMM path


def device_asm(obj):
  with tempfile.TemporaryDirectory() as d:
    fb, dev = os.path.join(d, "fb.bin"), os.path.join(d, "dev.o")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", obj, os.path.join(d, "x.o")], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"], check=True)
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", dev], check=True, capture_output=True, text=True).stdout


def vregs(text):
  """VGPR indices named in an operand string."""
  out = set()
  for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
    if m.group(3) is not None:
      out.add(int(m.group(3)))
    else:
      out.update(range(int(m.group(1)), int(m.group(2)) + 1))
  return out


def passes_of(mn):
  m = re.search(r"_(\d+x\d+x\d+)", mn)
  key = m.group(1) if m else ""
  for k, v in PASSES.items():
    if key.startswith(k):
      return v
  return 16          # unknown shape: the longest window


def check(obj):
  kernel, insts, bad = "?", [], []
  for line in device_asm(obj).splitlines():
    m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
    if m:
      insts.append(("label", m.group(1), "")); continue
    body = line.split("//")[0].strip()
    if not body or body.endswith(":"):
      continue
    parts = body.split(None, 1)
    insts.append((parts[0], parts[1] if len(parts) > 1 else "", body))
  n_mfma = 0
  for i, (mn, ops, body) in enumerate(insts):
    if mn == "label":
      kernel = ops; continue
    if not mn.startswith("v_mfma") and not mn.startswith("v_smfmac"):
      continue
    n_mfma += 1
    o = [x.strip() for x in ops.split(",")]
    P = passes_of(mn)
    dst = vregs(o[0])
    need_ab, need_c = (P + 3, P + 1) if is_xdl(mn) else (P + 2, P)
    states, j = 0, i + 1
    while j < len(insts) and states < need_ab:
      m2, o2, b2 = insts[j]
      if m2 == "label" or m2 in ("s_endpgm", "s_setpc_b64", "s_branch"):
        break
      if m2.startswith("v_mfma") or m2.startswith("v_smfmac"):
        q = [x.strip() for x in o2.split(",")]
        d2, a2, b2r, c2 = vregs(q[0]), vregs(q[1]), vregs(q[2]), vregs(q[3]) if len(q) > 3 else set()
        if (a2 | b2r) & dst:
          bad.append((kernel, body, insts[j][2], states))
        elif (d2 | c2) & dst and not (d2 == dst and c2 == dst) and states < need_c:
          bad.append((kernel, body, insts[j][2], states))
      elif vregs(o2) & dst:
        bad.append((kernel, body, insts[j][2], states))
      states += (int(o2, 0) + 1) if m2 == "s_nop" else 1
      j += 1
  return n_mfma, bad


if __name__ == "__main__":
  rc = 0
  for obj in sys.argv[1:]:
    n, bad = check(obj)
    print(f"{obj}: {n} MFMA instructions, {len(bad)} accesses to a destination inside its wait-state window")
    for k, a, b, s in bad[:20]:
      print(f"  {k[:60]}: `{a}` then `{b}` after {s} wait states")
    rc |= bool(bad)
  sys.exit(rc)
