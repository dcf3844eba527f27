#!/bin/bash
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igpz_amd/csrc tools/small_trace.hip -o build/small_trace 2>&1 | grep -E "error" ; build/small_trace "$@"
