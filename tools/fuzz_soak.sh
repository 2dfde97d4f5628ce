#!/bin/bash
# Soak runs of the randomised differential test (tests/test_gpu_fuzz.py) with fields of the drawn configuration FORCED -- run on the GPU box through gpurun:
#   tools/fuzz_soak.sh [seeds per run = 600] [first seed = 70000]      -> gpurun_out/soak_<name>.log (pass / fail counts, the failing seeds, their messages)
# Round 6 (seed 50160, 32-bit stage masks) was found by the first of these.  Combinations the library refuses by design (a pixel x receiver AND a
# pixel x transmit array outside fp32 / real weights / 'DAS') are kept out with "wpm": false.
N=${1:-600}; S0=${2:-70000}
mkdir -p gpurun_out
run() { # name offset override
  QDAS_FUZZ_OFFSET=$2 QDAS_FUZZ_SEEDS=$N QDAS_FUZZ_OVERRIDE="$3" timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -n 12 > gpurun_out/soak_$1.full 2>&1
  { grep -E "^FAILED|passed|failed" gpurun_out/soak_$1.full | tail -40; grep -E "^E  " gpurun_out/soak_$1.full | sort | uniq -c | sort -rn | head -10; } > gpurun_out/soak_$1.log
  rm -f gpurun_out/soak_$1.full
  tail -1 gpurun_out/soak_$1.log
}
run plain      $S0              ''
run jit        $((S0 + 1000))   '{"jit": true}'
run jit_half   $((S0 + 2000))   '{"jit": true, "prec": "halfT", "wpm": false}'
run jit_wm_wn  $((S0 + 3000))   '{"jit": true, "wm": true, "wn": true, "wpm": false}'
run jit_f4     $((S0 + 4000))   '{"jit": true, "F": 4}'
run jit_tz     $((S0 + 5000))   '{"jit": true, "tz": 16, "t0vec": true}'
run jit_ks     $((S0 + 6000))   '{"jit": true, "ks": 3, "t0vec": true}'
