#!/bin/bash
# rocprofv3 --pmc passes (each counter set in its own run, only --kernel-trace beside it): HBM traffic of the supervision path
# and of the full-resolution CRF, LDS counters of the mean-field filter.   usage: bash tools/gpu_pmc.sh outdir [what...]
OUT=${1:-gpurun_out/pmc}; shift
WHAT=${@:-"sup_fetch sup_write sup_lds fr_fetch fr_write"}
mkdir -p $OUT
ROOT=$PWD
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
run() {   # name counters script
  rm -rf /tmp/pmc_$1
  timeout 600 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pmc_$1 -o $1 -- python $ROOT/tools/$3 > $ROOT/$OUT/$1.log 2>&1
  f=$(find /tmp/pmc_$1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $ROOT/$OUT/$1_counter_collection.csv && echo "$1: $(wc -l < $f) rows"
}
for w in $WHAT; do
  case $w in
    sup_fetch) run sup_fetch "FETCH_SIZE" pmc_workload.py ;;
    sup_write) run sup_write "WRITE_SIZE" pmc_workload.py ;;
    sup_lds) run sup_lds "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" pmc_workload.py ;;
    fr_fetch) run fr_fetch "FETCH_SIZE" pmc_workload_fullres.py ;;
    fr_write) run fr_write "WRITE_SIZE" pmc_workload_fullres.py ;;
    fr_l2) run fr_l2 "TCC_HIT_sum TCC_MISS_sum" pmc_workload_fullres.py ;;
    train_mfma)
      rm -rf /tmp/pmc_train_mfma
      timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pmc_train_mfma -o train_mfma -- python $ROOT/bench.py --steps 6 --warmup 6 --no-fp32 --no-cpu-baseline --no-modes --no-profile > $ROOT/$OUT/train_mfma.log 2>&1
      f=$(find /tmp/pmc_train_mfma -name "*counter_collection.csv" | head -1)
      [ -n "$f" ] && cp $f $ROOT/$OUT/train_mfma_counter_collection.csv && echo "train_mfma: $(wc -l < $f) rows" ;;
    train_lds)
      rm -rf /tmp/pmc_train_lds
      timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_train_lds -o train_lds -- python $ROOT/bench.py --steps 4 --warmup 6 --no-fp32 --no-cpu-baseline --no-modes --no-profile > $ROOT/$OUT/train_lds.log 2>&1
      f=$(find /tmp/pmc_train_lds -name "*counter_collection.csv" | head -1)
      [ -n "$f" ] && cp $f $ROOT/$OUT/train_lds_counter_collection.csv && echo "train_lds: $(wc -l < $f) rows" ;;
  esac
done
cd $ROOT
