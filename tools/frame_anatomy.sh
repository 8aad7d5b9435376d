#!/bin/bash
# on the GPU box: per-frame kernel table of `bench.py ARGS` (rocprofv3 kernel trace) -> stdout.   usage: tools/frame_anatomy.sh TAG [rows] -- bench args
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; rows=${2:-24}; shift; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o r -- python3 $R/bench.py "$@" --no-cpu-baseline --soak 0 > /tmp/kt_$tag.json 2>/dev/null
python3 $R/tools/kernel_split.py /tmp/kt_$tag /tmp/kt_$tag.json $rows
