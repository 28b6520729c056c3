#!/bin/bash
# The randomised tile stress (tools/stress_tile.py) in both builds that have the 288 x 320 tile, under the allocator / serialisation
# variants the round-5 review named; logs under gpurun_out/<round>/ (copy what is to be judged to profiles/<round>/).
#   tools/stress_tile.sh r6 [launches per run]
R=${1:-r6}; N=${2:-2000}
OUT=gpurun_out/$R; mkdir -p "$OUT"
export MUDG_DEBUG_VARIANTS=1
rc=0
run() {  # name, env...
    local name=$1; shift
    echo "== $name" | tee -a "$OUT/stress_summary.txt"
    ( env "$@" timeout ${STRESS_TIMEOUT:-1500} python tools/stress_tile.py --launches "$N" --seed "$SEED" > "$OUT/stress_$name.log" 2>&1 ) || rc=1
    grep -E "DONE|FAIL|Error|error|abort" "$OUT/stress_$name.log" | head -20 | tee -a "$OUT/stress_summary.txt"
}
SEED=1 run bf16_default MUDG_OPERAND=bf16
SEED=2 run bf16_nocache MUDG_OPERAND=bf16 PYTORCH_NO_CUDA_MEMORY_CACHING=1
SEED=3 run bf16_serialize MUDG_OPERAND=bf16 AMD_SERIALIZE_KERNEL=3
SEED=5 run x3_default MUDG_OPERAND=bf16x3
SEED=6 run x3_nocache MUDG_OPERAND=bf16x3 PYTORCH_NO_CUDA_MEMORY_CACHING=1
# the 160 x 320 tile (round 6; 16-bit builds)
run160() {
    local name=$1; shift
    echo "== $name" | tee -a "$OUT/stress_summary.txt"
    ( env "$@" timeout ${STRESS_TIMEOUT:-1500} python tools/stress_tile.py --tile 160 --launches "$N" --seed "$SEED" > "$OUT/stress_$name.log" 2>&1 ) || rc=1
    grep -E "DONE|FAIL|Error|error|abort" "$OUT/stress_$name.log" | head -20 | tee -a "$OUT/stress_summary.txt"
}
SEED=7 run160 w160_bf16_default MUDG_OPERAND=bf16
SEED=8 run160 w160_bf16_nocache MUDG_OPERAND=bf16 PYTORCH_NO_CUDA_MEMORY_CACHING=1
SEED=9 run160 w160_bf16_serialize MUDG_OPERAND=bf16 AMD_SERIALIZE_KERNEL=3
exit $rc
