"""Turns the rocprofv3 --pmc passes of scripts/pmc_probe.py (gpurun_out/<dir>/pmc_*/pmc_counter_collection.csv) into
profiles/<tag>_pmc_summary.md and profiles/pmc_traffic.json (HBM-side bytes per launch of the dominant kernels).

Rows are keyed by (probe SEGMENT, kernel name): the probe starts every shape it launches with a one-workgroup marker
kernel and prints "SEG <i> <label>", and the dispatch sequence of each pass is cut at those markers.  (Round 4 keyed by
kernel NAME only and averaged wgrad_x3p_kernel<3,3,4> at level 1 (175k rows) and at level 2 (40k rows) into one row --
VERDICT round 4, "What's weak": the 553.7 MB / 343.6 us / 31.3 % row was the mean of two shapes.)

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of a 16-byte-per-lane read stream
(MI355X_MICROARCH.md "HBM"), so both are calibrated on the first kernel of the probe (eltwise add, known bytes)
before being applied to the conv kernels, whose loads are float4 per lane as well."""
import collections, csv, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out")
tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
WANT = ("spconv_mfma", "spconv16", "wgrad_mfma", "wgrad_x3t", "wgrad_x3p", "wgrad_slab", "wgrad_reduce", "sk_fixup", "x3_pack", "x3_image",
        "split_reduce")
MARK_MAX_GRID = 256  # the marker is ONE workgroup of eltwise_kernel<2>; the calibration launch of the same kernel is 2^17 workgroups


def short(name):
  return re.sub(r"void |pcmi::", "", name).split("(")[0]


def load(name):
  """{(segment, kernel): {counter: [(value, us), ...]}}; segment -1 = before the first marker (calibration)."""
  path = os.path.join(src, "pmc_%s" % name, "pmc_counter_collection.csv")
  by_dispatch = collections.OrderedDict()
  for r in csv.DictReader(open(path)):
    by_dispatch.setdefault(int(r["Dispatch_Id"]), []).append(r)
  d = collections.defaultdict(lambda: collections.defaultdict(list))
  seg = -1
  for did in sorted(by_dispatch):
    rows = by_dispatch[did]
    k = short(rows[0]["Kernel_Name"])
    if k.startswith("eltwise_kernel<2>") and int(rows[0]["Grid_Size"]) <= MARK_MAX_GRID:
      seg += 1
      continue
    for r in rows:
      d[(seg, k)][r["Counter_Name"]].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3))
  return d


def mean(xs):
  return sum(x[0] for x in xs) / len(xs)


F, W, M = load("FETCH_SIZE"), load("WRITE_SIZE"), load("SQ_VALU_MFMA_BUSY_CYCLES")
cal = [k for k in F if k[0] == -1 and k[1].startswith("eltwise_kernel<2>")][0]
cal_read, cal_write = 2 * (1 << 19) * 128 * 4, (1 << 19) * 128 * 4
f_scale = cal_read / (mean(F[cal]["FETCH_SIZE"]) * 1024)
w_scale = cal_write / (mean(W[cal]["WRITE_SIZE"]) * 1024)
algo, seg_label = {}, {-1: "calibration"}
for line in open(os.path.join(src, "pmc_FETCH_SIZE.log")):
  if line.startswith("ALGO"):
    m = dict(kv.split("=") for kv in line.split() if "=" in kv)
    algo[line.split()[2]] = {k: int(v) for k, v in m.items()}
  elif line.startswith("SEG "):
    seg_label[int(line.split()[1])] = line.split(None, 2)[2].strip()
rows, segments = [], collections.OrderedDict()
for key in sorted(F):
  seg, k = key
  if key != cal and not any(x in k for x in WANT):
    continue
  rd = mean(F[key]["FETCH_SIZE"]) * 1024 * f_scale
  wr = mean(W[key]["WRITE_SIZE"]) * 1024 * w_scale if key in W else float("nan")
  us = sum(x[1] for x in F[key]["FETCH_SIZE"]) / len(F[key]["FETCH_SIZE"])
  util = mops = float("nan")
  if key in M and "SQ_VALU_MFMA_BUSY_CYCLES" in M[key]:
    gui = mean(M[key]["GRBM_GUI_ACTIVE"]) / 8.0  # summed over the 8 XCDs
    util = mean(M[key]["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024.0 / gui  # per SIMD (256 CUs x 4)
    mops = mean(M[key]["SQ_INSTS_VALU_MFMA_MOPS_F32"]) * 512 * 1e-9
    if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in M[key]:  # the split-precision kernels: bf16 MFMA work (six products per fp32 product)
      mops += mean(M[key]["SQ_INSTS_VALU_MFMA_MOPS_BF16"]) * 512 * 1e-9
  label = seg_label.get(seg, "segment %d" % seg)
  rows.append((label, k, len(F[key]["FETCH_SIZE"]), us, rd * 1e-6, wr * 1e-6, util, mops))
  segments.setdefault(label, {})[k] = {"launches": len(F[key]["FETCH_SIZE"]), "us": round(us, 2), "read_bytes": rd, "written_bytes": wr,
                                       "mfma_busy": None if util != util else round(util, 4),
                                       "issued_gflop": None if mops != mops else round(mops, 3)}
md = ["# PMC summary (%s): rocprofv3 --kernel-trace --pmc <counter> -- python scripts/pmc_probe.py" % tag, "",
      "calibration kernel `eltwise_kernel<2>` (536.9 MB read, 268.4 MB written, float4 per lane): FETCH_SIZE x %.3f, WRITE_SIZE x %.3f"
      % (f_scale, w_scale), "",
      "algorithmic bytes / flops of the probed convs (SURVEY 8d formulae): " + json.dumps(algo), "",
      "Rows are per probe SEGMENT (one shape each) and kernel: the same kernel name on two shapes is two rows.", "",
      "| segment | kernel | launches | avg us (under counters) | read MB | written MB | MFMA busy (per SIMD) | issued GFLOP |", "|---|---|---|---|---|---|---|---|"]
for r in rows:
  md.append("| %s | `%s` | %d | %.1f | %.1f | %.1f | %.1f %% | %.2f |" % (r[0], r[1], r[2], r[3], r[4], r[5], 100 * r[6], r[7]))
md += ["", "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs); issued GFLOP = "
       "(SQ_INSTS_VALU_MFMA_MOPS_F32 + ..._BF16) x 512 (a split-precision kernel issues 6 bf16 products per fp32 product; compare with 6 x the "
       "algorithmic 2*M*Cin*Cout: the excess is MFMA work on absent neighbours)."]
# issued / algorithmic matrix work per (segment, matrix kernel) whose algorithmic flops the probe printed
md += ["", "| segment | kernel | issued GFLOP | 6 x algorithmic GFLOP | issued / algorithmic |", "|---|---|---|---|---|"]
for r in rows:
  a = None
  for name, v in algo.items():  # ALGO tag "level2:96->96" / "96->96": the segment label starts with the same "<tag><cin>-><cout>"
    if r[0].startswith(name + " "):
      a = v
  if a is None or r[7] != r[7] or r[7] <= 0 or not any(x in r[1] for x in ("spconv16x", "wgrad_x3")):
    continue
  per = a["flops"] * 1e-9 * 6
  md.append("| %s | `%s` | %.2f | %.2f | x%.3f |" % (r[0], r[1], r[7], per, r[7] / per))
  segments[r[0]][r[1]]["issued_over_algorithmic"] = round(r[7] / per, 4)
# LDS bank conflicts and wave-level stall attribution (optional fourth pass)
try:
  L = load("SQ_LDS_BANK_CONFLICT")
  md += ["", "| segment | kernel | LDS conflict cycles / LDS active cycles | waves: issuing | parked (s_waitcnt / barrier) | issue-stalled (matrix pipe / dependencies) |",
         "|---|---|---|---|---|---|"]
  for key in sorted(L):
    seg, k = key
    if not any(x in k for x in ("spconv16", "wgrad_mfma", "wgrad_x3t", "wgrad_x3p")):
      continue
    c = {name: mean(v) for name, v in L[key].items()}
    wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    md.append("| %s | `%s` | %.3f | %.1f %% | %.1f %% | %.1f %% |" % (seg_label.get(seg, seg), k, c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0),
                                                      100 * c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 100 * c.get("SQ_WAIT_ANY", 0.0) / wc,
                                                      100 * c.get("SQ_WAIT_INST_ANY", 0.0) / wc))
except (OSError, KeyError, IndexError) as e:
  md += ["", "(no LDS / stall pass: %s)" % e]
open(os.path.join(root, "profiles", "%s_pmc_summary.md" % tag), "w").write("\n".join(md) + "\n")
sys.path.insert(0, root)
from pointcontrast_amd.build import sources_digest  # noqa: E402  (the build these counters were collected on)
# bench.py's roofline.traffic: the LEVEL-1 96->96 segment (the first segment of the probe), by kernel name
first = next((lab for lab in segments if lab.startswith("96->96")), None)
traffic = {k: v["read_bytes"] + v["written_bytes"] for k, v in (segments.get(first) or {}).items()}
json.dump({"source": "%s_pmc_summary.md" % tag, "kernel_sources_sha16": sources_digest()[:16], "bytes_per_launch": traffic,
           "bytes_per_launch_segment": first, "segments": segments, "algorithmic": algo},
          open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print("\n".join(md))
