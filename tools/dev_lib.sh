#!/bin/bash
# Source me: builds (once) the -DLRX_DEV_KNOBS variant of liblrx.so and points the tools at it.  The environment switches of the A/B scripts
# (LRX_SS_FORCE, LRX_SS_MAX, LRX_SEARCH_FUSED, LRX_FUSED_PHASES, LRX_EMIT_PERSIST_MIN_BPC, LRX_GEMM_GM, LRX_MAXAGG_GM, LRX_EMIT_GM,
# LRX_ATTN_TILED, LRX_SEARCH_WIDE_MAX) exist only in that build; the shipping library reads no environment variable (tests/test_abi.py).
_lrx_root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
_lrx_dev="$_lrx_root/tools/exp/liblrx_dev.so"
if [ ! -f "$_lrx_dev" ] || [ -n "$(find "$_lrx_root/lightretriever_amd/csrc" "$_lrx_root/include" -newer "$_lrx_dev" -type f | head -1)" ]; then
  (cd "$_lrx_root" && python -m lightretriever_amd.build -DLRX_DEV_KNOBS --out="$_lrx_dev" > /dev/null)
fi
export LRX_LIB_DEV_VARIANT="${LRX_LIB_DEV_VARIANT:-$_lrx_dev}"
