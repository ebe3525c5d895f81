"""Turn one evidence run (gpurun_out/ev_*: bench line, rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE / SQ PMC passes of
the same bench.py command) into the tracked files under profiles/ for round RR.
usage: make_profiles.py <round tag, e.g. r02> [gpurun_out dir]"""
import csv, json, os, statistics, sys
from collections import defaultdict

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
out = "profiles"
bench = json.loads([l for l in open(f"{src}/ev_bench.json") if l.startswith("{")][-1])
prof_bench = json.loads([l for l in open(f"{src}/ev_stats.log", errors="ignore") if l.startswith("{")][-1])
steps = prof_bench["steps"] + prof_bench["warmup"]

# ---- 1. kernel stats
rows = list(csv.DictReader(open(f"{src}/ev_stats/a_kernel_stats.csv")))
# the timed + warm-up steps are not all there is (bench.py runs extra host-driven steps to time the BMU kernel): one optimizer launch = one step
adam_calls = [int(r["Calls"]) for r in rows if "adamw_kernel" in r["Name"]]
if adam_calls:
    steps = max(adam_calls)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(f"{out}/{tag}_bench_n1_kernel_stats_summary.txt", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps {prof_bench['steps']} --warmup {prof_bench['warmup']} --no-cpu-baseline\n")
    f.write(f"#   N=1, c3 workload (B=512), {steps} training steps in the trace (timed + warm-up + the extra steps that time the BMU kernel); bench line under the profiler {prof_bench['ms_per_step']:.2f} ms/step, "
            f"un-profiled run of the same build on the same box {bench['ms_per_step']:.2f} ms/step\n")
    f.write(f"# total kernel time {tot/1e6:.1f} ms; per-step = / {steps}.  Forward and backward run on two / three HIP streams: overlapping kernels\n"
            f"# share the GPU, their individual durations stretch and the per-step column sums to more than the wall-clock step.\n\n")
    for r in rows[:40]:
        f.write(f"{r['Name'][:96]:96s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  per-step {float(r['TotalDurationNs'])/1e6/steps:7.3f} ms  {float(r['Percentage']):5.1f}%\n")
with open(f"{out}/{tag}_bench_n1_kernel_stats.csv", "w") as f:
    f.write(open(f"{src}/ev_stats/a_kernel_stats.csv").read())

# ---- 2. HBM traffic
def load(path, counters):
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] in counters:
            per[r["Kernel_Name"]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return per
fetch = load(f"{src}/ev_pmc_f/a_counter_collection.csv", {"FETCH_SIZE"})
write = load(f"{src}/ev_pmc_w/a_counter_collection.csv", {"WRITE_SIZE"})
trows = []
for k in fetch:
    rd = 2.0 * statistics.median(fetch[k]["FETCH_SIZE"].values()) * 1024
    wr = statistics.median(write.get(k, {}).get("WRITE_SIZE", {0: 0.0}).values()) * 1024
    trows.append((len(fetch[k]["FETCH_SIZE"]) * (rd + wr), k, len(fetch[k]["FETCH_SIZE"]), rd, wr))
trows.sort(reverse=True)
bmu = [t for t in trows if "bmu_x3_planes_kernel" in t[1]] or [t for t in trows if "bmu_x3_kernel" in t[1]]
adam = [t for t in trows if "adamw_kernel" in t[1]]
with open(f"{out}/{tag}_bench_n1_pmc_hbm_traffic.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  rocprofv3 --kernel-trace --pmc WRITE_SIZE  (two separate passes)\n"
            "#   -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline      (N=1, c3 workload, B=512); table by tools/make_profiles.py\n"
            "# counter unit = KiB, summed over the 8 XCDs of a dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE\n"
            "# tallies the 128-B requests of wide coalesced reads at 64 B, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact.\n"
            "# Values are medians per launch.\n")
    if adam:
        f.write(f"# Calibration inside this run: adamw_kernel reads 4 x 100.4 MB = 401.6 MB algorithmic -> {adam[0][3]/1e6:.1f} MB measured; writes 3 x 100.4 = 301.2 MB -> {adam[0][4]/1e6:.1f} MB.\n")
    if bmu:
        alg = bench["roofline"]["algorithmic_bytes"]
        f.write(f"# BMU distance pass, contraction kernel ({bmu[0][1].split('(')[0]}): algorithmic {alg/1e6:.1f} MB (X 25.2 + W 78.6 + dist 3.3); measured {bmu[0][3]/1e6:.1f} MB read + {bmu[0][4]/1e6:.1f} MB written "
                f"(slabs of the L split + norm partials) = {(bmu[0][3]+bmu[0][4])/1e6:.1f} MB per launch = {(bmu[0][3]+bmu[0][4])/alg:.2f} x algorithmic.\n")
    f.write("\n")
    for _, k, n, rd, wr in trows[:40]:
        f.write(f"{k[:96]:96s} launches {n:4d}  read {rd/1e6:8.1f} MB  write {wr/1e6:8.1f} MB\n")
if bmu:
    json.dump({"kernel": bmu[0][1].split("(")[0].replace("vsom::", "").replace("void ", ""), "batch": 512, "read_bytes": round(bmu[0][3]), "write_bytes": round(bmu[0][4]),
               "source": f"profiles/{tag}_bench_n1_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py; read = 2 x FETCH_SIZE x 1024)"},
              open(f"{out}/{tag}_bmu_hbm_traffic.json", "w"), indent=1)

med = lambda d, c: statistics.median(d[c].values()) if d.get(c) else 0.0
# ---- 3. SQ instruction mix
sq = load(f"{src}/ev_pmc_s/a_counter_collection.csv", {"SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU"})
tq = load(f"{src}/ev_pmc_t/a_counter_collection.csv", {"SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"})
with open(f"{out}/{tag}_sq_instruction_mix.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_{VALU,MFMA,LDS,VMEM,SALU}  and  --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY\n"
            "#   SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (two passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline\n"
            "# per WAVE, medians per launch; cycle counters x 4 (the SQ counts quad-cycles).  wait_inst = issue stalls (MFMA pipe / dependencies / LDS),\n"
            "# wait_any = s_waitcnt + barriers; 'ldsconf' = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.\n\n")
    f.write(f"{'kernel':72s} {'waves':>7s} {'VALU':>7s} {'MFMA':>6s} {'LDS':>6s} {'VMEM':>6s} {'SALU':>6s} | {'cycles':>8s} {'active':>7s} {'wait_inst':>9s} {'wait_any':>8s} {'ldsconf':>7s}\n")
    keys = sorted(sq, key=lambda k: -med(sq[k], "SQ_INSTS_MFMA") * len(sq[k]["SQ_WAVES"]))
    for k in keys[:24]:
        w = med(sq[k], "SQ_WAVES")
        if w <= 0:
            continue
        t = tq.get(k, {})
        f.write(f"{k[:72]:72s} {w:7.0f} {med(sq[k],'SQ_INSTS_VALU')/w:7.0f} {med(sq[k],'SQ_INSTS_MFMA')/w:6.0f} {med(sq[k],'SQ_INSTS_LDS')/w:6.0f} "
                f"{med(sq[k],'SQ_INSTS_VMEM')/w:6.0f} {med(sq[k],'SQ_INSTS_SALU')/w:6.0f} | {4*med(t,'SQ_WAVE_CYCLES')/w:8.0f} {4*med(t,'SQ_ACTIVE_INST_ANY')/w:7.0f} "
                f"{4*med(t,'SQ_WAIT_INST_ANY')/w:9.0f} {4*med(t,'SQ_WAIT_ANY')/w:8.0f} {100*med(t,'SQ_LDS_BANK_CONFLICT')/max(med(t,'SQ_LDS_IDX_ACTIVE'),1):6.1f}%\n")

# ---- 3b. matrix-pipe busy time per launch
try:
    mb = load(f"{src}/ev_pmc_m/a_counter_collection.csv", {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"})
    dur = defaultdict(list)
    for r in csv.DictReader(open(f"{src}/ev_pmc_m/a_kernel_trace.csv")):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    NX = 8.0                                   # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs
    with open(f"{out}/{tag}_mfma_busy.txt", "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline\n"
                "# SQ_VALU_MFMA_BUSY_CYCLES = cycles in which a SIMD's matrix pipe is busy, summed over the 1024 SIMDs of the chip (32 per\n"
                "# v_mfma_f32_32x32x16_bf16, 32 per v_mfma_f32_16x16x4_f32): busy/SIMD = that / 1024.  GRBM_GUI_ACTIVE / 8 = GPU-active cycles of the\n"
                "# launch per XCD; it ticks at ~2.4 GHz whatever the shader clock does (the in-kernel s_memtime clock of the BMU kernel under\n"
                "# back-to-back load is 1.5-1.9 GHz, profiles/r03_bmu_planes_lab.txt), so 'pipe busy' = busy/SIMD / (GUI / 8) is a LOWER bound of\n"
                "# the fraction of shader cycles the matrix pipe is busy.  'pipe us' = busy/SIMD at 2.0 GHz.  Medians per launch; launch time from\n"
                "# the same pass (counters on).\n\n")
        f.write(f"{'kernel':84s} {'launch us':>9s} {'busy/SIMD kcyc':>14s} {'pipe us':>8s} {'GUI/8 kcyc':>10s} {'pipe busy':>9s}\n")
        for k in sorted(mb, key=lambda k: -med(mb[k], "SQ_VALU_MFMA_BUSY_CYCLES") * len(mb[k]["SQ_VALU_MFMA_BUSY_CYCLES"]))[:24]:
            busy = med(mb[k], "SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0
            gui = med(mb[k], "GRBM_GUI_ACTIVE") / NX
            us = statistics.median(dur[k]) / 1e3 if dur.get(k) else 0.0
            if busy <= 0 or gui <= 0:
                continue
            f.write(f"{k[:84]:84s} {us:9.1f} {busy/1e3:14.1f} {busy/2.0e3:8.1f} {gui/1e3:10.1f} {busy/gui:9.2f}\n")
except OSError:
    pass

# ---- 4. per-kernel roofline table
with open(f"{out}/{tag}_kernel_roofline_table.txt", "w") as f:
    f.write("# Per-kernel roofline table, c3 shapes (T = 33280 tokens, E = 192), each kernel ALONE (tools/layer_gemms.py; bench.py 'roofline' /\n"
            "# 'secondary').  Peaks (MI355X_MICROARCH.md): HBM 8.0 TB/s; f32 MFMA 157.3 TF; bf16 MFMA 2500 TF -> 416.7 TF f32-equivalent for the\n"
            "# six-product split engine (forward GEMMs), 833.3 TF for the three-product engine (gradient GEMMs of the default mode, BMU contraction).\n"
            "# frac = achieved / peak of the binding roofline.\n\n")
    for l in open(f"{src}/ev_layer_gemms.log"):
        if "us" in l and "TF" in l:
            parts = l.split()
            tf = float(parts[parts.index("TF") - 1])
            prod = int(parts[-1]) if parts[-2] == "products" else 6
            f.write(l.rstrip() + f"   frac(mfma {2500.0/prod:.1f} TF) {tf/(2500.0/prod):.3f}\n")
        elif l.startswith("sum"):
            f.write(l)
    r = bench["roofline"]; s2 = bench["secondary"]
    if r.get("traffic") is None and bmu:
        r["traffic"] = round(bmu[0][3] + bmu[0][4])        # the bench ran before this round's traffic file existed
    f.write(f"\nBMU distance pass   {r['kernel'][:60]}...  {r['avg_launch_ms']*1e3:.1f} us  {r['achieved']:.1f} TF f32-eq  frac {r['frac']:.3f} of {r['peak']} TF;  "
            f"HBM view {r['hbm_view']['achieved_GBps']:.0f} GB/s = {r['hbm_view']['frac']:.3f} of 8 TB/s; traffic {r['traffic']}\n")
    for k, v in s2.items():
        a = v.get("achieved_TFLOPs", v.get("achieved_f32_equiv_TFLOPs"))
        f.write(f"{k:24s} {v['ms']*1e3:7.1f} us  {a:7.1f} TF  frac {v['frac']:.3f} of {v['peak_TFLOPs']} TF  ({v['hbm_GBps']} GB/s)\n")
    f.write(f"\nstep: {bench['ms_per_step']} ms = {bench['value']} images/s;  cpu_baseline: {json.dumps(bench.get('cpu_baseline'))}\n")
open(f"{out}/{tag}_bench_n1.json", "w").write(json.dumps(bench) + "\n")
print("wrote profiles for", tag)
