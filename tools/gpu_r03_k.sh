#!/bin/bash
set -u
P=gpurun_out/r03; mkdir -p $P; export TMPDIR=/tmp
SECONDS_BUDGET=100 timeout 400 python tests/soak/soak_r03.py > $P/soak_r03.txt 2>&1; echo "soak_r03 rc=$? $(tail -1 $P/soak_r03.txt | cut -c1-300)"
SECONDS_BUDGET=60 timeout 300 python tests/soak/soak.py > $P/soak_chaos_codec.txt 2>&1; echo "soak rc=$? $(tail -1 $P/soak_chaos_codec.txt | cut -c1-200)"
SECONDS_BUDGET=60 timeout 300 python tests/soak/step_stress.py > $P/step_stress.txt 2>&1; echo "step_stress rc=$? $(tail -1 $P/step_stress.txt | cut -c1-300)"
