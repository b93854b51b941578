# A second build of the library under another name, for A/B timing of two builds inside ONE gpurun call
# (distributed-groth16_amd/lib.py loads $DG16_LIB when it is set):
#   bash tools/build_variant.sh noasm "-DDG29_NO_ASM_MAD"   -> distributed-groth16_amd/libdg16_noasm.so
# The sources are copied to /tmp/dg16_variant_<name> and built there: the objects of the tree are not touched.
set -e
name=$1; xflags=$2; shift 2   # further arguments go to make (e.g. OUTLINE_GROUPS=bls12_381_g2)
root=$(cd "$(dirname "$0")/.." && pwd)
work=/tmp/dg16_variant_$name
rm -rf $work; mkdir -p $work/distributed-groth16_amd $work/include
cp -r $root/distributed-groth16_amd/csrc $work/distributed-groth16_amd/
cp $root/include/*.h $work/include/
mkdir -p $work/tools && cp $root/tools/check_agpr_file.py $root/tools/true.py $work/tools/
rm -f $work/distributed-groth16_amd/csrc/*.o $work/distributed-groth16_amd/csrc/*.usage.txt
make -s -j"$(nproc)" -C $work/distributed-groth16_amd/csrc XFLAGS="$xflags" "$@"
cp $work/distributed-groth16_amd/libdg16.so $root/distributed-groth16_amd/libdg16_$name.so
echo "built distributed-groth16_amd/libdg16_$name.so with $xflags"
