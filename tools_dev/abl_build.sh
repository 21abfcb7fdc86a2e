#!/bin/bash
# dev helper: build a variant library with extra -D flags for the LP=8 translation unit
#   tools_dev/abl_build.sh NAME -DFOO=1 ...   -> waldo_amd/lib/abl/NAME.so
set -e
name=$1; shift
mkdir -p waldo_amd/lib/abl /tmp/abl_$name
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Iinclude "$@" \
  -c waldo_amd/csrc/warp_composite_lp8.hip -o /tmp/abl_$name/lp8.o
objs=$(ls waldo_amd/csrc/_obj/*.o | grep -v warp_composite_lp8.o)
hipcc -shared -fPIC --offload-arch=gfx950 -o waldo_amd/lib/abl/$name.so $objs /tmp/abl_$name/lp8.o
echo built waldo_amd/lib/abl/$name.so
