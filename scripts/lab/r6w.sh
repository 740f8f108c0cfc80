#!/bin/bash
# Round 6, GPU call W: the new 2-rank dynamic-scale test on one GPU (recorded programs, flag exchange as a host callback) + the agent tests.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_agent.py -x -q -m gpu > $O/pytest_agent.txt 2>&1; tail -15 $O/pytest_agent.txt
