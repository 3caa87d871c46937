#!/bin/bash
# tools/gpu_steps.sh <outdir> <<'STEPS' ... : runs one GPU step per input line "name|seconds|command", each under its own
# time limit, output to <outdir>/<name>.log.  A step that ends in a time limit, abort, segfault or kill (124, 137, 134,
# 139) ends the script: nothing more is started on the GPU in that call.  Ordinary failures (a failing test) do not.
set -uo pipefail
out=$1; mkdir -p "$out"
while IFS='|' read -r name secs cmd; do
  [ -z "${name:-}" ] && continue
  echo "== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "   rc=$rc: $(tail -n 1 "$out/$name.log" | cut -c1-300)"
  case $rc in 124|137|134|139) echo "!! $name ended with $rc: stopping here"; exit $rc;; esac
done
exit 0
