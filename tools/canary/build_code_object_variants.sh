#!/bin/bash
# Does it matter whether victim and aggressor of the first-read effect come from one code object or two?  Builds, for first_read_repro_dl.cpp:
#   /tmp/libvictim_small.so   the GEMV sources alone (api + rowops, guard compiled out)          - aggressor then from /tmp/libsynth.so
#   /tmp/libcombo_two_tu.so   GEMV sources + synthetic aggressor, two translation units           - one library, two code objects
#   /tmp/libcombo_one_tu.so   the same two sources #included into ONE translation unit            - one library, ONE code object
set -e
R=${1:-$(pwd)}
cd $R
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iladcast_amd/csrc -Wno-unused-value -DLDC_AB_BUILD -DLDC_LS_NO_FIRST_READ"
hipcc $F -shared ladcast_amd/csrc/api.hip ladcast_amd/csrc/rowops.hip -o /tmp/libvictim_small.so
hipcc $F -shared ladcast_amd/csrc/api.hip ladcast_amd/csrc/rowops.hip tools/canary/synthetic_aggressor.hip -o /tmp/libcombo_two_tu.so
cat > /tmp/_combo_one_tu.hip <<'EOT'
#include "rowops.hip"
#define main synth_main_unused
#include "synthetic_aggressor.hip"
EOT
hipcc $F -Itools/canary -shared ladcast_amd/csrc/api.hip /tmp/_combo_one_tu.hip -o /tmp/libcombo_one_tu.so
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -shared -fPIC tools/canary/synthetic_aggressor.hip -o /tmp/libsynth.so
hipcc -O2 tools/canary/first_read_repro_dl.cpp -o /tmp/repro_dl -ldl
