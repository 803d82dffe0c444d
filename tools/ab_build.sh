#!/bin/bash
# Builds libqmri.so of another commit into tools/ab/libqmri_<name>.so (git-ignored, travels with gpurun) so that two builds can be
# timed on the same box in one call:  QMRI_LIBQMRI=tools/ab/libqmri_<name>.so python bench.py ...
#   tools/ab_build.sh <commit> <name>
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP="$(mktemp -d)"
git -C "$ROOT" archive "$1" qmri_pnp_recon_poc_amd/csrc include | tar -x -C "$TMP"
make -C "$TMP/qmri_pnp_recon_poc_amd/csrc" -s -j8
mkdir -p "$ROOT/tools/ab"
cp "$TMP/qmri_pnp_recon_poc_amd/libqmri.so" "$ROOT/tools/ab/libqmri_$2.so"
rm -rf "$TMP"
echo "built tools/ab/libqmri_$2.so from $1"
