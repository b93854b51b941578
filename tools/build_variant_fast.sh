# Like build_variant.sh, but starts from the tree's OBJECTS and rebuilds only the named ones with the extra flags:
#   bash tools/build_variant_fast.sh ls128 "-DDG16_G2_LOCKSTEP" msm_bls12_381_g2.o msm_bls12_377_g2.o
# (for switches that change the code of a few translation units only; the tree must be built)
set -e
name=$1; xflags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
work=/tmp/dg16_variant_$name
rm -rf $work; mkdir -p $work/distributed-groth16_amd $work/include
cp -a $root/distributed-groth16_amd/csrc $work/distributed-groth16_amd/
cp -a $root/include/*.h $work/include/
mkdir -p $work/tools && cp $root/tools/check_agpr_file.py $root/tools/true.py $work/tools/
cd $work/distributed-groth16_amd/csrc
for o in "$@"; do rm -f $o; done
make -s -j"$(nproc)" XFLAGS="$xflags" ../libdg16.so
cp $work/distributed-groth16_amd/libdg16.so $root/distributed-groth16_amd/libdg16_$name.so
echo "built distributed-groth16_amd/libdg16_$name.so with $xflags ($*)"
