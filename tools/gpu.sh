#!/bin/bash
# tools/gpu.sh <timeout_s> '<command>' : one gpurun call, output to gpurun_out/last.log
/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" > gpurun_out/last.log 2>&1
tail -${3:-40} gpurun_out/last.log
