#!/bin/bash
# ONE parameterised driver for every GPU call of a round (rounds 3-4 kept a shell script per call: archived in profiles/gpu_calls_r03_r04.md).
#   scripts/gpu_call.sh <name> [timeout_s] < commands.sh
# runs the commands read from stdin on a fresh MI355X box through gpurun, from the root of the repository snapshot, with
# O=gpurun_out/<name> created and exported; whatever the commands write under $O comes back into gpurun_out/<name>/ here.  The commands of
# every call of round 5 are kept beside their results: gpurun_out/<name>/commands.sh is copied to profiles/r05_calls/<name>.sh by hand for
# the calls whose results are cited.
set -e
name=${1:?usage: scripts/gpu_call.sh <name> [timeout_s] < commands}
limit=${2:-1800}
body=$(cat)
mkdir -p gpurun_out/$name
printf '%s\n' "$body" > gpurun_out/$name/commands.sh
exec /usr/local/graft/bin/gpurun --timeout "$limit" -- "cd \$GRAFT_REPO_ROOT; export O=gpurun_out/$name PYTHONUNBUFFERED=1 TMPDIR=/tmp; mkdir -p \$O; $body"
