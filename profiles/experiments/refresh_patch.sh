#!/bin/bash
# Re-bases profiles/experiments/instrumentation.patch (or another patch given as $1) onto the csrc sources of the working tree after they moved: a three-way merge per
# file (git merge-file) of  base = csrc of the last commit the patch applied to (default HEAD),  ours = base + patch,  theirs = the working tree;  the difference between
# the working tree and the merge result is written back as the new patch.  A conflict stops the script and names the file.
#   bash profiles/experiments/refresh_patch.sh [patch file] [base commit]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
PATCH=${1:-$ROOT/profiles/experiments/instrumentation.patch}
BASE=${2:-HEAD}
if [ -n "$RESOLVED" ]; then
    WORK=$RESOLVED
    if grep -l '^<<<<<<< \|^>>>>>>> ' "$WORK"/c/csrc/* 2>/dev/null; then echo "conflict markers left in the files above"; exit 1; fi
    for f in "$ROOT"/raytracing-in-one-weekend_amd/csrc/*.h "$ROOT"/raytracing-in-one-weekend_amd/csrc/*.hip "$ROOT"/raytracing-in-one-weekend_amd/csrc/*.cpp "$ROOT"/raytracing-in-one-weekend_amd/csrc/Makefile; do cp "$f" "$WORK/a/csrc/"; done
    find "$WORK/c" \( -name "*.orig" -o -name "*.rej" \) -delete
    (cd "$WORK" && diff -ru a/csrc c/csrc > new.patch) || true
    cp "$WORK/new.patch" "$PATCH"; rm -rf "$WORK"; echo "refreshed $PATCH"; exit 0
fi
WORK=$(mktemp -d)
mkdir -p "$WORK/base" "$WORK/c" "$WORK/a/csrc"
(cd "$ROOT" && git archive "$BASE" raytracing-in-one-weekend_amd/csrc | tar -x -C "$WORK")
cp -r "$WORK/raytracing-in-one-weekend_amd/csrc" "$WORK/base/csrc"
mv "$WORK/raytracing-in-one-weekend_amd/csrc" "$WORK/c/csrc"
(cd "$WORK/c" && patch -p1 -s < "$PATCH")
for f in "$ROOT"/raytracing-in-one-weekend_amd/csrc/*.h "$ROOT"/raytracing-in-one-weekend_amd/csrc/*.hip "$ROOT"/raytracing-in-one-weekend_amd/csrc/*.cpp "$ROOT"/raytracing-in-one-weekend_amd/csrc/Makefile; do
    n=$(basename "$f")
    cp "$f" "$WORK/a/csrc/$n"
    if [ -f "$WORK/c/csrc/$n" ] && [ -f "$WORK/base/csrc/$n" ]; then
        git merge-file -q "$WORK/c/csrc/$n" "$WORK/base/csrc/$n" "$f" || CONFLICTS="$CONFLICTS $n"
    elif [ ! -f "$WORK/c/csrc/$n" ]; then
        cp "$f" "$WORK/c/csrc/$n"
    fi
done
if [ -n "$CONFLICTS" ]; then
    # both sides changed the same lines: resolve the markers in the files named below, then run   RESOLVED=$WORK bash profiles/experiments/refresh_patch.sh
    echo "conflicts in:$CONFLICTS - edit them under $WORK/c/csrc, then: RESOLVED=$WORK bash $0 $PATCH"; exit 1
fi
find "$WORK/c" \( -name "*.orig" -o -name "*.rej" \) -delete
(cd "$WORK" && diff -ru a/csrc c/csrc > new.patch) || true
cp "$WORK/new.patch" "$PATCH"
rm -rf "$WORK"
echo "refreshed $PATCH"
