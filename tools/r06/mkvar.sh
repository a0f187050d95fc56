# usage: mkvar.sh name "-DFLAG ..." file.hip : compile one translation unit with extra flags and link a variant library
name=$1; flags=$2; src=$3
cd /root/repo
obj=/tmp/var_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function $flags -c ffwm_amd/csrc/$src.hip -o $obj || exit 1
objs=$(ls ffwm_amd/build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_variants/$name.so $objs $obj && echo built $name
