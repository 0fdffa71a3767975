#!/bin/bash
# Round 5, the two laggards of the loops: the last fine sweep of the V-cycle fused with r.z (EPI_JACOBI_DOT on the
# row-tile kernel) and update_r.  Builds lab variants of the library (one object recompiled with a -D each), then -- on the GPU box, all in
# ONE session so that the box is the same -- runs bench.py under rocprofv3 --kernel-trace with each of them and prints the kernels' medians.
#   build (here, no GPU):      bash conjugategradient_amd/tools/epi_variants.sh build
#   measure (GPU box):         bash conjugategradient_amd/tools/epi_variants.sh run OUTDIR
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/conjugategradient_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fvisibility=hidden"
OBJS="runtime ops solver comm kernels_spmv kernels_rows kernels_dcsr kernels_tiled kernels_mg spectrum"
if [ "$1" = build ]; then
  make -s -j8 -C $CS || exit 1
  for v in 1 2 3; do
    /opt/rocm/bin/hipcc $FLAGS -DMGCG_EPI_VARIANT=$v -c $CS/kernels_rowtile.hip -o /tmp/rowtile_epi$v.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/conjugategradient_amd/tools/libMgcgGpu_epi$v.so $(for o in $OBJS kernels_blas1; do echo $CS/$o.o; done) /tmp/rowtile_epi$v.o -ldl -lpthread || exit 1
  done
  for v in 1 2 3; do
    /opt/rocm/bin/hipcc $FLAGS -DMGCG_R_VARIANT=$v -c $CS/kernels_blas1.hip -o /tmp/blas1_r$v.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/conjugategradient_amd/tools/libMgcgGpu_r$v.so $(for o in $OBJS kernels_rowtile; do echo $CS/$o.o; done) /tmp/blas1_r$v.o -ldl -lpthread || exit 1
  done
  ls -la $ROOT/conjugategradient_amd/tools/libMgcgGpu_epi*.so $ROOT/conjugategradient_amd/tools/libMgcgGpu_r*.so
  exit 0
fi
OUT=$2; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
one() { local name=$1 lib=$2 solver=$3 pat=$4
  (cd /tmp && env MGCG_LIB_PATH=$ROOT/$lib rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/trace_$name -- python3 $ROOT/bench.py --solver $solver --steps 30 --warmup 5 --no-extras --no-cpu-baseline) > $OUT/$name.json 2> $OUT/$name.err
  echo "== $name" >> $OUT/medians.log
  python3 conjugategradient_amd/tools/trace_kernel_medians.py $OUT/trace_$name "$pat" >> $OUT/medians.log
  rm -rf $OUT/trace_$name; }
: > $OUT/medians.log
for round in 1 2; do
  one base_mgcg_$round conjugategradient_amd/libMgcgGpu.so mgcg "spmv_rowtile_kernel<6"
  for v in 1 2 3; do one epi${v}_mgcg_$round conjugategradient_amd/tools/libMgcgGpu_epi$v.so mgcg "spmv_rowtile_kernel<6"; done
  one base_cg_$round conjugategradient_amd/libMgcgGpu.so cg "update_r"
  for v in 1 2 3; do one r${v}_cg_$round conjugategradient_amd/tools/libMgcgGpu_r$v.so cg "update_r"; done
done
cat $OUT/medians.log
