#!/bin/bash
# round 6, call c: the reset + one iteration sequence of the failing test with finite checks after every body
O=gpurun_out/r8c; mkdir -p $O
timeout 120 python scripts/dbg/mixed_graphs.py "d_fb,d_opt,g_opt,r1_fb" lp reset > $O/mixed_reset.txt 2>&1
timeout 120 python scripts/dbg/mixed_graphs.py "g_fb,d_fb,d_opt,g_opt,r1_fb" lp reset > $O/all_reset.txt 2>&1
grep -c after $O/*.txt
