#!/bin/bash
# Round 6, GPU call Y: the box's power limits as rocm-smi reports them (read only), beside DESIGN 3.1's power paragraph.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6y; mkdir -p $O
( rocm-smi --showmaxpower; rocm-smi --showpower; rocm-smi --showperflevel; rocm-smi --showclocks; rocm-smi --showvoltage; rocm-smi --showtemp; rocm-smi --showpowerprofile ) > $O/smi_limits.txt 2>&1
rocm-smi -a > $O/smi_all.txt 2>&1
grep -i "power\|cap\|watt" $O/smi_all.txt | head -30
