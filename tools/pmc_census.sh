#!/bin/bash
# Per-kernel PMC census of one bench step (round 5): for every kernel of the step, how busy the texture-data path and the LDS array are
# and how many bytes an L2 read request carries -- the comparison that found the dominant kernel's half-line fetches (finding 52).
#   bash tools/pmc_census.sh   (through gpurun, from the repo root)  ->  gpurun_out/pmc_census.txt
REPO=$(pwd); OUT=$REPO/gpurun_out/${CENSUS_OUT:-pmc_census.txt}; mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
# CENSUS_EXTRA="--precision fp16x3 --batch 16": the default precision's step (round 6)
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --stack3d 0 --engine2d 0 --latency 0 --fine-boundaries 0 --fp32-mode 0 ${CENSUS_EXTRA:-}"
run() { name=$1; shift; rm -rf /tmp/cen_$name; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/cen_$name -o p -- python3 $REPO/bench.py $ARGS > /tmp/cen_$name.log 2>&1; echo "pass $name rc=$?"; }
run a GRBM_GUI_ACTIVE TD_TD_BUSY_sum TA_TA_BUSY_sum
run b TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
run d FETCH_SIZE
run e WRITE_SIZE
timeout 120 python3 $REPO/tools/pmc_census.py /tmp/cen_a /tmp/cen_b /tmp/cen_c /tmp/cen_d /tmp/cen_e > $OUT 2>&1
cat $OUT
