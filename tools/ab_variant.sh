#!/bin/bash
# tools/ab_variant.sh <file.hip> <tag> [git-rev | -D...]: a second library libdir_hip_<tag>.so in which ONE translation unit is built from
# another revision of its source (git show <rev>:...) or with extra -D flags; select it with DIR_HIP_LIBRARY for same-box A/B runs.
# Development only: the variant libraries are git-ignored.
set -e
cd "$(dirname "$0")/../details-in-recommendation_amd"
f=$1; tag=$2; shift 2
python3 build.py > /dev/null
src=csrc/$f.hip; defs=""
for a in "$@"; do
  case $a in
    -D*) defs="$defs $a";;
    *) git show "$a:details-in-recommendation_amd/csrc/$f.hip" > csrc/_build/${f}_$tag.hip; src=csrc/_build/${f}_$tag.hip;;
  esac
done
extra=$(python3 -c "
import build
print(' '.join(build.EXTRA_FLAGS.get('$f.hip', [])))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip -Wall -Wno-unused-function -Icsrc -I../include $extra $defs \
  -c $src -o csrc/_build/variant_${f}_$tag.o 2>&1 | grep -v "not a recognized" || true
objs=$(ls csrc/_build/*.o | grep -v "/$f.o\|/variant_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libdir_hip_$tag.so $objs csrc/_build/variant_${f}_$tag.o
echo "built details-in-recommendation_amd/libdir_hip_$tag.so"
