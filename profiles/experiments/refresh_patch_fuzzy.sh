#!/bin/bash
# Re-bases profiles/experiments/instrumentation.patch onto the WORKING TREE's csrc when only the context moved: applies the patch with fuzz to a copy and writes the
# difference back.  (refresh_patch.sh does a three-way merge against a commit, which needs the committed patch to apply to the committed sources.)  A rejected hunk stops it.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
PATCH=${1:-$ROOT/profiles/experiments/instrumentation.patch}
W=$(mktemp -d)
mkdir -p $W/a $W/c
cp -r $ROOT/raytracing-in-one-weekend_amd/csrc $W/a/csrc
rm -rf $W/a/csrc/build $W/a/csrc/*.so
cp -r $W/a/csrc $W/c/csrc
(cd $W/c && patch -p1 -F3 -s < $PATCH)
if find $W/c -name "*.rej" | grep -q .; then echo "rejected hunks under $W/c"; exit 1; fi
find $W/c -name "*.orig" -delete
(cd $W && diff -ru a/csrc c/csrc > new.patch) || true
cp $W/new.patch $PATCH
rm -rf $W
echo "refreshed $PATCH"
