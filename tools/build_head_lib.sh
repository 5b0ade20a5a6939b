#!/bin/bash
# Build the committed HEAD of this repository into meanflow_audio_codec_amd/csrc/libmfc_head.so (git-ignored), for
# same-box A/B runs against the working tree: MFC_LIB=$PWD/meanflow_audio_codec_amd/csrc/libmfc_head.so python tools/...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/mfc_head.XXXXXX)
git -C "$ROOT" worktree add --detach "$W" HEAD > /dev/null 2>&1
(cd "$W" && python -m meanflow_audio_codec_amd._build --force > /dev/null)
cp "$W/meanflow_audio_codec_amd/csrc/libmfc.so" "$ROOT/meanflow_audio_codec_amd/csrc/libmfc_head.so"
git -C "$ROOT" worktree remove --force "$W"
echo "$ROOT/meanflow_audio_codec_amd/csrc/libmfc_head.so"
