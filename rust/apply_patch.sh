#!/bin/sh
# Makes the patched copy of ndarray-interp 0.6.0 that rust/ndarray-interp-hip/Cargo.toml depends on.
#   rust/apply_patch.sh /path/to/a/checkout/of/ndarray-interp-0.6.0
# Copies the checkout to rust/vendor/ndarray-interp and applies rust/patches/ndarray-interp-0.6.0-batched-hook.patch
# (one defaulted batched method per strategy trait; interp_array_into routed through it).  Nothing is downloaded.
set -eu
src=${1:?usage: rust/apply_patch.sh <ndarray-interp 0.6.0 checkout>}
here=$(cd "$(dirname "$0")" && pwd)
dst="$here/vendor/ndarray-interp"
rm -rf "$dst"
mkdir -p "$here/vendor"
cp -R "$src" "$dst"
rm -rf "$dst/.git" "$dst/target"
(cd "$dst" && patch -p1 < "$here/patches/ndarray-interp-0.6.0-batched-hook.patch")
echo "patched copy in $dst"
