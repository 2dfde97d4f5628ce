#!/bin/bash
# Regenerate EVERY measured line quoted in README.md / DESIGN.md at HEAD, on one GPU box, in one go:
#   tools/profile_all.sh [round]        -> gpurun_out/profiles_<round>/   (copy what is to be judged into profiles/<round>/)
# bench lines (one JSON per workload / switch), rocprofv3 summaries (kernel trace + separate PMC passes: tools/profile.sh) of the
# headline and the other BASELINE configurations, and the register tables of the prebuilt kernels AND of the hiprtc builds the
# benches ran (their code objects are in this run's private cache directory).
set -u
R=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/profiles_$R
mkdir -p $OUT
export QDAS_CACHE_DIR=$OUT/jit_cache
export QDAS_JIT_SPEC_LOG=$OUT/jit_specs.txt      # every hiprtc build of this run as a complete spec: sort -u into tests/jit_kernels.txt (rebuilt without a device by tests/test_jit.py)
cd $REPO
b() { # name args...
  local name=$1; shift
  python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || echo "bench $name FAILED" >> $OUT/errors.txt
  [ -s $OUT/bench_$name.err ] || rm -f $OUT/bench_$name.err
}
# ---- headline (the driver's command) and the other BASELINE configurations, with cpu_baseline + parity_check + live traffic
b c3_default
b c2 --workload c2 --steps 50 --warmup 5
b c5 --workload c5 --steps 50 --warmup 5
b c1 --workload c1 --steps 200 --warmup 20
# ---- what the special modes are worth (no CPU leg, no counter passes)
Q="--no-cpu --no-traffic --no-general"
b c3_no_fold --no-fold $Q --steps 10
b c3_frames4 --frames 4 --no-cpu --no-general --steps 5
QDAS_NO_MIRROR=1 b c3_no_mirror $Q --steps 10
QDAS_NO_MIRROR=1 b c3_no_mirror_no_fold --no-fold $Q --steps 10
QDAS_NO_MIRROR=1 b c2_no_mirror --workload c2 $Q --steps 50
QDAS_NO_MIRROR=1 b c1_no_mirror --workload c1 $Q --steps 200
b c3_general --no-reciprocal $Q --steps 10
QDAS_NO_MIRROR=1 b c3_general_no_mirror --no-reciprocal $Q --steps 10
b c3_fp16 --prec halfT $Q --steps 10
b c3_fmod --fmod 5e6 $Q --steps 10
b c3_window --window-apod $Q --steps 10
b c3_fnumber1.5 --rx-apod fnumber:1.5 --no-cpu --no-general --steps 10
b c3_fp16_fnumber1.5 --prec halfT --rx-apod fnumber:1.5 $Q --steps 10
QDAS_NO_MIRROR=1 b c5_no_mirror --workload c5 $Q --steps 50
b c5_single --workload c5 --prec single $Q --steps 50
QDAS_NO_STAGE_SHAPE=1 b c5_r05_shape --workload c5 $Q --steps 50
b c2_fp16 --workload c2 --prec halfT $Q --steps 50
b c2_double --workload c2 --prec double $Q --steps 10
b c2_double_fmod --workload c2 --prec double --fmod 5e6 $Q --steps 10
b c2_window --workload c2 --window-apod $Q --steps 20
b pw9 --workload pw9 $Q --steps 100
b c1f --workload c1f $Q --steps 100
b c1_multiline --workload c1 --tx-apod multiline $Q --steps 100
b c1_multiline_acceptance --workload c1 --tx-apod multiline --rx-apod acceptance:30 --rx-apod-array $Q --steps 100
# ---- rocprofv3: kernel trace + PMC passes
for w in c3 c2 c5; do
  bash tools/profile.sh ${R}_$w --workload $w > /dev/null 2>&1
  cp gpurun_out/prof_${R}_$w/summary.txt $OUT/rocprofv3_summary_$w.txt 2>/dev/null
  cp gpurun_out/prof_${R}_$w/traffic.json $OUT/traffic_$w.json 2>/dev/null
  f=$(ls gpurun_out/prof_${R}_$w/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/rocprofv3_kernel_stats_$w.csv
done
bash tools/profile.sh ${R}_c3_general --workload c3 --no-reciprocal > /dev/null 2>&1
cp gpurun_out/prof_${R}_c3_general/summary.txt $OUT/rocprofv3_summary_c3_general.txt 2>/dev/null
# ---- streams of frames, the general (non-fused) kernels
python tools/frames_bench.py c3 6 2>/dev/null | grep -v amdgpu.ids > $OUT/frames_bench.txt; python tools/frames_bench.py c2 12 >> $OUT/frames_bench.txt 2>/dev/null; python tools/frames_bench.py c5 12 >> $OUT/frames_bench.txt 2>/dev/null; python tools/frames_bench.py c1 12 >> $OUT/frames_bench.txt 2>/dev/null
python tools/general_time.py 2>/dev/null | grep -v Warn > $OUT/general_time.txt
python tools/convd_fft_time.py 2>/dev/null | grep -v amdgpu.ids > $OUT/convd_fft_time.txt
{ timeout 600 python tools/greens_time.py --stages 2>/dev/null | grep -v amdgpu.ids; echo '# QDAS_GREENS_NO_TRAINS=1 (round 3 kernel):'; QDAS_GREENS_NO_TRAINS=1 python tools/greens_time.py 2>/dev/null | grep -v amdgpu.ids; echo '# general_time.py greens lines, trains / QDAS_GREENS_NO_TRAINS=1:'; grep -i '^greens' $OUT/general_time.txt; QDAS_GREENS_NO_TRAINS=1 python tools/general_time.py 2>/dev/null | grep -i '^greens'; } > $OUT/greens_time.txt
# per-kernel durations of the same script (kernel-trace only): the launches behind every line of general_time.txt
(cd /tmp && TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_general -o g -- python $REPO/tools/general_time.py > /dev/null 2>&1)
python - <<PYEOF > $OUT/rocprofv3_summary_general.txt
import csv, glob
fs = glob.glob("$OUT/trace_general/**/*kernel_stats.csv", recursive=True)
print("rocprofv3 --kernel-trace --stats -- python tools/general_time.py   (kernel_stats.csv; durations in ns)")
if fs:
    rows = list(csv.DictReader(open(fs[0])))
    print(f"{'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}  kernel")
    for r in rows[:40]:
        print(f"{r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.1f} {float(r['MinNs']) / 1e3:10.1f} {float(r['MaxNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}  {r['Name'][:150]}")
else:
    print("no kernel_stats.csv")
PYEOF
rm -rf $OUT/trace_general
# ---- one rank's slab on one GPU (NOT a scaling curve), the issue-cost microbenchmark, the ablation of the headline kernel
python tools/slab_scaling.py c3 2>/dev/null | grep -v amdgpu.ids > $OUT/slab_kernel_times_c3.txt
python tools/issue_rates.py 2>/dev/null | grep -v amdgpu.ids > $OUT/issue_rates.txt
mkdir -p tools/scratch && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/microbench_src/f16mix.hip -o tools/scratch/f16mix 2>/dev/null && tools/scratch/f16mix > $OUT/f16mix.txt 2>&1
bash tools/dma_ab.sh 2 > $OUT/dma_ab_c3_raw.txt 2>&1
{ echo '# tools/abl_sweep.sh 2 "--workload c5": hiprtc builds of BASELINE C5 with QDAS_ABL bits (1 no LDS-DMA, 16 no barrier, 2048 no priority staircase), kernel ms'; tools/abl_sweep.sh 2 "--workload c5" - QDAS_ABL=1 QDAS_ABL=16 QDAS_ABL=17 QDAS_ABL=2048; } > $OUT/ablation_c5_raw.txt 2>&1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/microbench_src/issue.hip -o /tmp/qdas_issue 2>/dev/null && /tmp/qdas_issue > $OUT/issue_costs.txt 2>&1
{ echo "# tools/abl_sweep.sh: hiprtc builds with QDAS_JIT_DEFINES=QDAS_ABL=<bits> (tile_hooks.h), interleaved rounds, kernel ms (fold pass included).  Bits: 1 no LDS-DMA, 4 taps from registers,";
  echo "# 8 trivial weights, 16 no end-of-stage wait / barrier, 256 plain instead of software-pipelined pair loop, 1024 no late DMA, 2048 no priority staircase.";
  echo "# --- headline (reciprocity-folded frame + lateral-mirror mode)";
  tools/abl_sweep.sh 2 "" - QDAS_ABL=1 QDAS_ABL=4 QDAS_ABL=8 QDAS_ABL=12 QDAS_ABL=16 QDAS_ABL=256 QDAS_ABL=1024 QDAS_ABL=2048;
  echo "# --- the unfolded reciprocal + mirror kernel of round 3 (--no-fold)";
  tools/abl_sweep.sh 2 "--no-fold" - QDAS_ABL=1 QDAS_ABL=4 QDAS_ABL=8 QDAS_ABL=12 QDAS_ABL=16 QDAS_ABL=2048; } > $OUT/ablation_c3.txt 2>&1
# ---- PCIe-inclusive: host-resident frames through the C ABI, and the int16 RF -> hilbert -> band-pass -> DAS chain for a stream
python tools/host_frames.py > $OUT/host_frames.txt 2>/dev/null
python tools/pipeline_bench.py c3 12 64 > $OUT/pipeline_bench.txt 2>/dev/null; python tools/pipeline_bench.py c2 12 64 >> $OUT/pipeline_bench.txt 2>/dev/null
python tools/c1_chain.py 1000 2>/dev/null | grep -v Warn > $OUT/c1_chain.txt; python tools/c1_chain.py 3 2>/dev/null | grep -v Warn >> $OUT/c1_chain.txt
# ---- registers: prebuilt library, and the hiprtc builds of this run
python tools/kernel_regs.py qups_amd/libqdas.so > $OUT/kernel_regs.txt 2>&1
python tools/kernel_regs.py $QDAS_CACHE_DIR > $OUT/kernel_regs_hiprtc.txt 2>&1
sort -u $OUT/jit_specs.txt > $OUT/jit_kernels.txt 2>/dev/null
python - <<PY > $OUT/summary.txt
import glob, json, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(f))
        r = d["roofline"]; pc = d.get("parity_check", {})
        print(f"{os.path.basename(f)[6:-5]:28s} ms_per_step {d['ms_per_step']:8.3f}  kernel_ms {r['kernel_ms']:8.3f}  prebuilt {d.get('prebuilt_kernel_ms')}  general {d.get('general_ms_per_step')}  "
              f"Mpixel/s {d['value']:9.2f}  parity {pc.get('rel_err')} ok={pc.get('ok')}  traffic {r.get('traffic')}  exec_frac {r.get('pairs_executed_frac')} valu_exec {r.get('valu_frac_executed')}  {d['config']['kernel_name']}")
    except Exception as ex:
        print(os.path.basename(f), "unreadable:", ex)
PY
cat $OUT/summary.txt
