#!/bin/bash
# gpurun wrapper: creates gpurun_out/r04 on the box before the command.  Usage: tools/grun.sh <timeout> '<command>'
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout $t -- "mkdir -p gpurun_out/r04 && { $*; }"
