"""Register / LDS footprint of every kernel of libpcmi, from hipcc's own resource remarks (no GPU needed).

  python scripts/kernel_resources.py [source.hip ...]  [-D...]     -> table on stdout

Why it matters here ("what comes next" 3): one wgrad_x3p_kernel<3,3,4> workgroup per CU leaves 52 VGPRs per
SIMD lane and 12 KiB of LDS -- a kernel of the backward chain that needs more runs on the 32 CUs the weight-gradient
launch leaves free, one that fits runs on all 256."""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcontrast_amd import build as B


def demangle(names):
  try:
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()
  except OSError:
    return names


def resources(src, extra=()):
  cmd = [B._hipcc()] + B.FLAGS + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.CSRC, src), "-o", "/dev/null"]
  err = subprocess.run(cmd, capture_output=True, text=True).stderr
  rows, cur = [], None
  for line in err.splitlines():
    m = re.search(r"remark: \s*([A-Za-z ]+?)(?: \[[^\]]*\])?:\s*(\S+)", line)
    if not m:
      continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
      cur = {"name": v}
      rows.append(cur)
    elif cur is not None:
      cur[k] = v
  names = demangle([r["name"] for r in rows])
  for r, n in zip(rows, names):
    r["name"] = re.sub(r"\(.*", "", n).replace("void ", "").replace("pcmi::", "").replace("(anonymous namespace)::", "")
  return rows


def main():
  args = [a for a in sys.argv[1:] if not a.startswith("-")]
  extra = [a for a in sys.argv[1:] if a.startswith("-")]
  print("%-14s %-52s %5s %5s %7s %5s %4s" % ("source", "kernel", "VGPR", "AGPR", "LDS B", "SGPR", "occ"))
  for src in (args or B.SOURCES):
    for r in resources(src, extra):
      print("%-14s %-52s %5s %5s %7s %5s %4s" % (src.replace(".hip", ""), r["name"][:52], r.get("VGPRs", "?"), r.get("AGPRs", "?"),
                                                 r.get("LDS Size", "?"), r.get("SGPRs", "?"), r.get("Occupancy", "?")))


if __name__ == "__main__":
  main()
