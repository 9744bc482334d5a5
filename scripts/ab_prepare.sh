#!/bin/bash
# exp = working tree (+ extra -D flags given here), base = HEAD.  Afterwards the in-tree libfi_hip.so is the
# BASE build: run `python -m feature_intertwiner_amd.build --force` before anything else uses it.
set -e
cd "$(dirname "$0")/.."
scripts/ab_build.sh "$@"
git stash -q
python -m feature_intertwiner_amd.build --force > /dev/null
git stash pop -q
echo "base = HEAD in libfi_hip.so, exp = working tree in libfi_hip_exp.so"
