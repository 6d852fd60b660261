#!/bin/bash
# full GPU suite
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -x -q > $O/c22_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/c22_tests.log
