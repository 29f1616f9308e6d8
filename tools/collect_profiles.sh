#!/bin/bash
# Collects the round's judged evidence on the GPU box into gpurun_out/profiles_${ROUND:-r05}/ (copy to profiles/ afterwards):
#   kernels_seq.txt     rocprofv3 --kernel-trace --stats of sequential pairs (un-overlapped kernel durations)
#   kernels_3streams.txt the default bench schedule (three streams)
#   pmc.txt             PMC counters (separate passes, FETCH_SIZE / WRITE_SIZE apart) of every kernel of the hot path
#   bench_line.json     the default bench.py line
#   train_step_kernels.txt  kernel summary of the full-size training step (tools/train_step_bench.py)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/profiles_${ROUND:-r05}
rm -rf $OUT; mkdir -p $OUT
SEQ="bench.py --steps 5 --warmup 1 --kernel-reps 2 --no-cpu-baseline --no-train-record --no-sub-records --no-pipeline --windows 1"
rocprofv3 --kernel-trace --stats -d $OUT/trace_seq -- python $SEQ > $OUT/seq.log 2>&1
python tools/prof_summary.py $OUT/trace_seq $OUT/kernels_seq.txt "$SEQ (sequential pairs on one stream: un-overlapped kernel durations; 6 hot-path passes + whole-network forwards + isolated conv64 launches)" > /dev/null 2>&1
PIPE="bench.py --steps 10 --warmup 2 --kernel-reps 2 --no-cpu-baseline --no-train-record --no-sub-records --windows 1"
rocprofv3 --kernel-trace --stats -d $OUT/trace_pipe -- python $PIPE > $OUT/pipe.log 2>&1
python tools/prof_summary.py $OUT/trace_pipe $OUT/kernels_3streams.txt "$PIPE (default schedule: pairs dealt to three HIP streams; durations overlap)" > /dev/null 2>&1
i=0
for C in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$i -- python bench.py --steps 2 --warmup 1 --kernel-reps 2 --no-cpu-baseline --no-train-record --no-sub-records --no-pipeline --windows 1 > $OUT/pmc_$i.log 2>&1
done
python tools/pmc_summary.py $OUT > $OUT/pmc.txt 2>&1
python bench.py > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log > $OUT/bench_line.json
python tools/train_step_bench.py > $OUT/train_step.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace_train -- python tools/train_step_bench.py > $OUT/train_prof.log 2>&1
python tools/prof_summary.py $OUT/trace_train $OUT/train_step_kernels.txt "tools/train_step_bench.py: 3 full-size training steps (960x540, D=192, one GPU, SubpixelCrossEntropy); un-profiled timing: $(tail -1 $OUT/train_step.log)" bygrid > /dev/null 2>&1
rm -rf $OUT/trace_seq $OUT/trace_pipe $OUT/trace_train $OUT/pmc_[0-9]   # raw databases stay on the box
ls -la $OUT
