"""Turns the rocprofv3 --pmc passes of scripts/pmc_probe.py (gpurun_out/pmc_*/pmc_counter_collection.csv) into
profiles/<tag>_pmc_summary.md and profiles/pmc_traffic.json (HBM-side bytes per launch of the dominant kernels).

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of a 16-byte-per-lane read stream
(MI355X_MICROARCH.md "HBM"), so both are calibrated on the first kernel of the probe (eltwise add, known bytes)
before being applied to the conv kernels, whose loads are float4 per lane as well."""
import collections, csv, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out")
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"


def load(name):
  d = collections.defaultdict(lambda: collections.defaultdict(list))
  path = os.path.join(src, "pmc_%s" % name, "pmc_counter_collection.csv")
  for r in csv.DictReader(open(path)):
    k = re.sub(r"void |pcmi::", "", r["Kernel_Name"])
    d[k][r["Counter_Name"]].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3))
  return d


def mean(xs):
  return sum(x[0] for x in xs) / len(xs)


F, W, M = load("FETCH_SIZE"), load("WRITE_SIZE"), load("SQ_VALU_MFMA_BUSY_CYCLES")
cal = [k for k in F if k.startswith("eltwise_kernel<2>")][0]
cal_read, cal_write = 2 * (1 << 19) * 128 * 4, (1 << 19) * 128 * 4
f_scale = cal_read / (mean(F[cal]["FETCH_SIZE"]) * 1024)
w_scale = cal_write / (mean(W[cal]["WRITE_SIZE"]) * 1024)
algo = {}
for line in open(os.path.join(src, "pmc_FETCH_SIZE.log")):
  if line.startswith("ALGO"):
    m = dict(kv.split("=") for kv in line.split()[3:])
    algo[line.split()[2]] = {k: int(v) for k, v in m.items()}
rows, traffic = [], {}
for k in F:
  if not any(x in k for x in ("spconv_mfma", "spconv16", "wgrad_mfma", "wgrad_x3t", "wgrad_x3p", "wgrad_slab", "sk_fixup", "x3_pack", "eltwise_kernel<2>")):
    continue
  rd = mean(F[k]["FETCH_SIZE"]) * 1024 * f_scale
  wr = mean(W[k]["WRITE_SIZE"]) * 1024 * w_scale if k in W else float("nan")
  us = sum(x[1] for x in F[k]["FETCH_SIZE"]) / len(F[k]["FETCH_SIZE"])
  util = mops = float("nan")
  if k in M and "SQ_VALU_MFMA_BUSY_CYCLES" in M[k]:
    gui = mean(M[k]["GRBM_GUI_ACTIVE"]) / 8.0  # summed over the 8 XCDs
    util = mean(M[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024.0 / gui  # per SIMD (256 CUs x 4)
    mops = mean(M[k]["SQ_INSTS_VALU_MFMA_MOPS_F32"]) * 512 * 1e-9
    if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in M[k]:  # the split-precision kernel: bf16 MFMA work (six products per fp32 product)
      mops += mean(M[k]["SQ_INSTS_VALU_MFMA_MOPS_BF16"]) * 512 * 1e-9
  rows.append((k.split("(")[0], len(F[k]["FETCH_SIZE"]), us, rd * 1e-6, wr * 1e-6, util, mops))
  traffic[k.split("(")[0]] = rd + wr
md = ["# PMC summary (%s): rocprofv3 --kernel-trace --pmc <counter> -- python scripts/pmc_probe.py" % tag, "",
      "calibration kernel `eltwise_kernel<2>` (536.9 MB read, 268.4 MB written, float4 per lane): FETCH_SIZE x %.3f, WRITE_SIZE x %.3f"
      % (f_scale, w_scale), "",
      "algorithmic bytes of the probed convs (SURVEY 8d formulae): " + json.dumps(algo), "",
      "| kernel | launches | avg us (under counters) | read MB | written MB | MFMA busy (per SIMD) | issued GFLOP |", "|---|---|---|---|---|---|---|"]
for r in rows:
  md.append("| `%s` | %d | %.1f | %.1f | %.1f | %.1f %% | %.2f |" % (r[0], r[1], r[2], r[3], r[4], 100 * r[5], r[6]))
md += ["", "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs); issued GFLOP = "
       "(SQ_INSTS_VALU_MFMA_MOPS_F32 + ..._BF16) x 512 (the split-precision kernel issues 6 bf16 products per fp32 product; compare with the algorithmic 2*M*Cin*Cout: the excess is MFMA work on absent neighbours)."]
# LDS bank conflicts and wave-level stall attribution (optional fourth pass)
try:
  L = load("SQ_LDS_BANK_CONFLICT")
  md += ["", "| kernel | LDS conflict cycles / LDS active cycles | waves: issuing | parked (s_waitcnt / barrier) | issue-stalled (matrix pipe / dependencies) |",
         "|---|---|---|---|---|"]
  for k in L:
    if not any(x in k for x in ("spconv16", "wgrad_mfma", "wgrad_x3t", "wgrad_x3p")):
      continue
    c = {name: mean(v) for name, v in L[k].items()}
    wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    md.append("| `%s` | %.3f | %.1f %% | %.1f %% | %.1f %% |" % (k.split("(")[0], c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0),
                                                      100 * c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 100 * c.get("SQ_WAIT_ANY", 0.0) / wc,
                                                      100 * c.get("SQ_WAIT_INST_ANY", 0.0) / wc))
except (OSError, KeyError, IndexError) as e:
  md += ["", "(no LDS / stall pass: %s)" % e]
open(os.path.join(root, "profiles", "%s_pmc_summary.md" % tag), "w").write("\n".join(md) + "\n")
sys.path.insert(0, root)
from pointcontrast_amd.build import sources_digest  # noqa: E402  (the build these counters were collected on)
json.dump({"source": "%s_pmc_summary.md" % tag, "kernel_sources_sha16": sources_digest()[:16], "bytes_per_launch": traffic,
           "algorithmic": algo},
          open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print("\n".join(md))
