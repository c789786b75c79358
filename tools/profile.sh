#!/bin/bash
# rocprofv3 passes on the GPU box (run through gpurun).  Every pass is wrapped in `timeout`; outputs go to
# gpurun_out/prof_<tag>/ (CSV + the rocpd database), summaries are copied to profiles/ by hand.
#   tools/profile.sh trace <tag> <command...>                 kernel trace + per-kernel stats
#   tools/profile.sh pmc <tag> "<counter counter ...>" <command...>   one counter pass (no tracing: gpurun refuses the mix)
mode=$1; tag=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
rm -rf $out            # a pass directory only ever holds its own pass (a round-1 leftover was committed as round 3 once)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
if [ "$mode" = trace ]; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- "$@" > $out/stdout.log 2>&1 < /dev/null
else
  ctrs=$1; shift
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $out -o $tag -- "$@" > $out/stdout.log 2>&1 < /dev/null
fi
echo "rc=$? files:"; find $out -type f | head -20
