#!/bin/bash
# Run from the container: stamps the tree that is about to be shipped (.git_head = HEAD's short hash; .git_dirty = 1 when the
# working tree differs from it -- the box has no .git), then hands the command to gpurun.
#   tools/gpu.sh <timeout seconds> '<command>'
set -u
cd "$(dirname "$0")/.."
git rev-parse --short HEAD > .git_head
if [ -n "$(git status --porcelain --untracked-files=no)" ]; then echo 1 > .git_dirty; else echo 0 > .git_dirty; fi
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
