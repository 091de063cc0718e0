#!/bin/bash
# Runs on the GPU box (gpurun): a list of steps "name|seconds|command", one after the other.  A step that fails with an
# ordinary exit code is recorded and the next one runs; a step that is KILLED at its limit (124 / 137) ends the session -
# nothing else is started on a GPU that may be wedged.  Output of every step goes to gpurun_out/<session>/<name>.log.
#   tools/gpu_steps.sh <session> "name|seconds|command" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
S=gpurun_out/$1; shift
mkdir -p "$S"
for step in "$@"; do
  name=${step%%|*}; rest=${step#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  t0=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "$S/$name.log" 2> "$S/$name.err"
  rc=$?
  echo "[$name] rc=$rc $(( $(date +%s) - t0 ))s"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping the session"; exit 1; fi
done
exit 0
