#!/bin/bash
# round 6, call m: the whole GPU suite on the current tree + smoke
O=gpurun_out/r8m; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_all.txt 2>&1; echo "all gpu tests rc=$?"; tail -5 $O/tests_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
