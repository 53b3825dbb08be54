#!/bin/bash
# round 6, call b: which mixed graph / eager combination of the step bodies produces NaN (test g_eager_d_graph of r8a)
O=gpurun_out/r8b; mkdir -p $O
for m in "g_fb,d_fb,r1_fb,g_opt,d_opt" "d_fb,d_opt,g_opt,r1_fb" "d_fb" "g_opt" "d_opt" "r1_fb" "g_fb"; do
  timeout 120 python scripts/dbg/mixed_graphs.py "$m" lp >> $O/mixed.txt 2>&1 || echo "rc=$? for $m" >> $O/mixed.txt
done
timeout 120 python scripts/dbg/mixed_graphs.py "d_fb,d_opt,g_opt,r1_fb" >> $O/mixed.txt 2>&1
grep -c "it 8" $O/mixed.txt
