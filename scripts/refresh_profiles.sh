#!/bin/bash
# Re-measure everything profiles/ holds for the current tree (run on the GPU box through gpurun; writes gpurun_out/$1):
#   default bench line, rocprofv3 kernel stats of the same command, PMC FETCH / WRITE passes of scripts/pmc_probe.py,
#   generator-forward and 128x1024 bench lines, micro-benchmark tables.
set -u
OUT=${1:-refresh}
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$OUT
mkdir -p $D
cd $R
python bench.py --steps 20 --warmup 5 > $D/bench_default.log 2>&1
python bench.py --workload gfwd --batch-per-gpu 32 --steps 50 --warmup 10 > $D/bench_gfwd.log 2>&1
python bench.py --res 128x1024 --batch-per-gpu 32 --steps 10 --warmup 4 > $D/bench_128x1024_bf16.log 2>&1
python bench.py --res 128x1024 --batch-per-gpu 32 --steps 10 --warmup 4 --dtype fp8 > $D/bench_128x1024_fp8.log 2>&1
(python scripts/mb_modup.py; python scripts/mb_conv_fp8.py; python scripts/mb_linear_x3.py; python scripts/mb_pewgrad2.py; python scripts/mb_conv.py; python scripts/mb_conv_s2.py; python scripts/mb_conv_x3.py; python scripts/mb_conv8.py) 2>&1 | grep -v amdgpu.ids > $D/microbench_tables.txt
# one rank on RCCL: the collectives of the N > 1 path (flat all-reduces, the tail exchange) with nobody to wait for
DGV2_DIST_WORLD1=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $D/bench_one_rank_rccl.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -- python3 $R/bench.py --steps 20 --warmup 5 > $D/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/pmc_f -- python3 $R/scripts/pmc_probe.py > $D/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/pmc_w -- python3 $R/scripts/pmc_probe.py > $D/pmc_w.log 2>&1
cd $R
python scripts/pmc_collect.py $D/pmc_f $D/pmc_w > $D/pmc.json 2> $D/pmc_collect.err
f=$(find $D/prof -name "*kernel_stats.csv" | head -1); cp $f $D/kernel_stats.csv; python scripts/prof_buckets.py $D/kernel_stats.csv > $D/buckets.txt
