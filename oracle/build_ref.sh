#!/usr/bin/env bash
# build_ref.sh -- build the REFERENCE rasterizer itself for gfx950 into oracle/_ref/.
#
# TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (the dev container);
# the GPU box uses the prebuilt oracle/_ref/libgsr_ref.so that travels with the snapshot.
#
# Recipe: the reference's own cuda_rasterizer/*.{cu,h} are read where they lie, translated
# by the IMAGE's hipify-perl (/opt/rocm/bin) into a throw-away temp dir, and compiled with
# hipcc together with our C-ABI driver (oracle/ref_driver.cpp).  No stand-in headers or
# libraries are written: CUB maps to the image's hipCUB, cooperative_groups to HIP's, glm is
# the reference's vendored copy.  The only source edits are mechanical: normalise the
# `<< <grid, block >> >` launch spelling to `<<<...>>>`, drop three #include lines hipify
# cannot map and that are redundant under HIP (device_launch_parameters.h,
# cooperative_groups/reduce.h -- unused, cub/device/device_radix_sort.cuh -- covered by
# hipcub.hpp), and map CUDA's __trap() to __builtin_trap() on the command line.
# Only the .so lands in oracle/_ref/ (git-ignored, NOT gpurun-ignored); no reference source
# or translated source is kept.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${GSR_REFERENCE_ROOT:-/root/reference}/gaussian_splatting/submodules/diff-gaussian-rasterization"
OUT="$HERE/_ref"
if [ ! -d "$REF/cuda_rasterizer" ]; then
  echo "build_ref.sh: reference sources not found at $REF (expected on the GPU box) -- skipping" >&2
  exit 0
fi
TMP="$(mktemp -d /tmp/gsr_refbuild.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
for f in "$REF"/cuda_rasterizer/*.cu "$REF"/cuda_rasterizer/*.h; do
  /opt/rocm/bin/hipify-perl "$f" > "$TMP/$(basename "$f")" 2>/dev/null
done
sed -i -e '/#include ""/d' -e '/cooperative_groups\/reduce.h/d' -e '/cub\/device\/device_radix_sort.cuh/d' "$TMP"/*.cu "$TMP"/*.h
sed -i -e 's/<< *</<<</g' -e 's/>> *>/>>>/g' "$TMP"/*.cu
mkdir -p "$OUT"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -D__trap=__builtin_trap -I"$REF/third_party/glm" -I"$TMP")
for f in forward backward rasterizer_impl; do
  /opt/rocm/bin/hipcc "${FLAGS[@]}" -x hip -c "$TMP/$f.cu" -o "$TMP/$f.o" &
done
/opt/rocm/bin/hipcc "${FLAGS[@]}" -x hip -c "$HERE/ref_driver.cpp" -o "$TMP/ref_driver.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgsr_ref.so" "$TMP"/forward.o "$TMP"/backward.o "$TMP"/rasterizer_impl.o "$TMP"/ref_driver.o
echo "built $OUT/libgsr_ref.so"
