set -e
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -o a -- python3 $ROOT/bench.py --steps 2 --warmup 1 --depth 4 --no-cpu-baseline > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/b -o b -- python3 $ROOT/bench.py --steps 2 --warmup 1 --depth 4 --no-cpu-baseline > /dev/null 2> $OUT/b.err
ls $OUT/a $OUT/b
