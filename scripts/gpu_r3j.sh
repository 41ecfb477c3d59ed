#!/bin/bash
mkdir -p gpurun_out/r3j; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bl -o bl -- python /root/repo/scripts/probes/blaslt_names.py > /root/repo/gpurun_out/r3j/run.log 2>&1
f=$(find /tmp/bl -name "*kernel_stats.csv" | head -1)
cp "$f" /root/repo/gpurun_out/r3j/blaslt_kernel_stats.csv
cut -c1-400 "$f" | head -20
