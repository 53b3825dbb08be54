#!/bin/bash
# Re-measure everything profiles/ holds for the current tree (run on the GPU box through gpurun; writes gpurun_out/$1):
#   default bench line, the kernel trace of the same command + its per-body listing (scripts/step_listing.py: text and the
#   instance JSON bench.py ranks `roofline` by), PMC FETCH / WRITE passes of scripts/pmc_probe.py, generator-forward,
#   128x1024 and one-rank-RCCL bench lines (+ the kernel trace of the latter), micro-benchmark tables.
# Every step runs under its own timeout (a hung step must not eat the call).
set -u
OUT=${1:-refresh}
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$OUT
# a FRESH directory per refresh: the listing below is cut from the one trace this call writes, stamped by the traced
# process itself (bench.py --stamp; step_listing.py checks the trace's pid against the stamp and copies its hashes)
rm -rf $D
mkdir -p $D
cd $R
line() { python -c "
import json,sys
try:
    d=json.loads([l for l in open('$D/$1.log') if l.startswith('{')][-1]); print('$1', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), (d.get('roofline') or {}).get('frac'), (d.get('extra') or {}).get('ms_plain_iteration'), (d.get('extra') or {}).get('ms_r1_iteration'))
except Exception as e: print('$1', 'ERR', e)"; }
timeout 900 python bench.py --steps 20 --warmup 5 > $D/bench_default.log 2> $D/bench_default.err; line bench_default
timeout 300 python bench.py --workload gfwd --batch-per-gpu 32 --steps 50 --warmup 10 > $D/bench_gfwd.log 2>&1; line bench_gfwd
timeout 600 python bench.py --res 128x1024 --batch-per-gpu 32 --steps 10 --warmup 4 --no-cpu-baseline > $D/bench_128x1024_bf16.log 2>&1; line bench_128x1024_bf16
timeout 600 python bench.py --res 128x1024 --batch-per-gpu 32 --steps 10 --warmup 4 --no-cpu-baseline --dtype fp8 > $D/bench_128x1024_fp8.log 2>&1; line bench_128x1024_fp8
(for s in mb_modup mb_conv_fp8 mb_linear_x3 mb_dgrad_actbwd mb_pewgrad2 mb_conv mb_conv_s2 mb_conv_x3 mb_conv8 mb_s2d mb_midgemm mb_rng; do timeout 300 python scripts/$s.py; done) 2>&1 | grep -v amdgpu.ids > $D/microbench_tables.txt
# one rank on RCCL: the collectives of the N > 1 path (captured reductions on a side stream, the tail exchange) with nobody to wait for
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $D/bench_noextra.log 2>/dev/null; line bench_noextra
DGV2_DIST_WORLD1=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $D/bench_one_rank_rccl.log 2> $D/bench_one_rank_rccl.err; line bench_one_rank_rccl
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --stamp $D/prof_stamp.json > $D/prof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/pmc_f -- python3 $R/scripts/pmc_probe.py > $D/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/pmc_w -- python3 $R/scripts/pmc_probe.py > $D/pmc_w.log 2>&1
export DGV2_DIST_WORLD1=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D/prof_rccl -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --stamp $D/prof_rccl_stamp.json > $D/prof_rccl.log 2>&1
unset DGV2_DIST_WORLD1 RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT
cd $R
python scripts/pmc_collect.py $D/pmc_f $D/pmc_w > $D/pmc.json 2> $D/pmc_collect.err
f=$(ls -t $(find $D/prof -name "*kernel_trace.csv") | head -1); python scripts/step_listing.py $f --full --json $D/step_instances.json --stamp $D/prof_stamp.json --kernel "conv_x3_kernelILi1|conv_pipe_kernel.*Li32ELi1ELi4ELi|conv3x3_strip|modconv_up_kernel" > $D/step_listing.txt; head -36 $D/step_listing.txt
f=$(ls -t $(find $D/prof -name "*kernel_stats.csv") | head -1); cp $f $D/bench_kernel_stats.csv   # same process as the listing
f=$(ls -t $(find $D/prof_rccl -name "*kernel_trace.csv") | head -1); python scripts/step_listing.py $f --full --kernel "nccl|rccl|oneRank|Reduce" > $D/one_rank_rccl_listing.txt; head -6 $D/one_rank_rccl_listing.txt
