#!/bin/bash
# One evidence run on the GPU box (everything tools/make_profiles.py reads).  From the repo root:
#   gpurun --timeout 1100 -- 'bash tools/collect_evidence.sh'   then   python3 tools/make_profiles.py rNN
# rocprofv3: kernel trace + stats in one pass, every --pmc group in a pass of its own (never with other trace domains).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --steps 50 --warmup 10 > "$OUT/ev_bench.json" 2> "$OUT/ev_bench.err"
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ev_stats" -o a -- python3 "$ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/ev_stats.log" 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/ev_pmc_f" -o a -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/ev_pmc_f.log" 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/ev_pmc_w" -o a -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/ev_pmc_w.log" 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d "$OUT/ev_pmc_s" -o a -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/ev_pmc_s.log" 2>&1
echo "sq insts done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/ev_pmc_t" -o a -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/ev_pmc_t.log" 2>&1
echo "sq cycles done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/ev_pmc_m" -o a -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/ev_pmc_m.log" 2>&1
echo "mfma busy done"
python3 "$ROOT/tools/layer_gemms.py" > "$OUT/ev_layer_gemms.log" 2>&1
echo "layer gemms done"
python3 "$ROOT/tools/host_vs_gpu.py" > "$OUT/ev_host_vs_gpu.log" 2>&1
echo "host vs gpu done"
python3 "$ROOT/tools/accuracy_modes.py" > "$OUT/ev_accuracy_modes.log" 2>&1
echo "accuracy done"
(cd "$ROOT" && ./lab/bmu_planes_lab > "$OUT/ev_bmu_planes_lab.log" 2>&1) || true
echo "bmu lab done"
# keep what travels back small: the per-dispatch traces are large, the tables are built from these files only
find "$OUT"/ev_* -name "*_kernel_trace.csv" -size +20M -delete 2>/dev/null || true
du -sh "$OUT"/ev_* | tail -12
