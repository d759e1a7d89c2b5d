#!/bin/bash
# builds the AQL probe (tools/aql_probe): device code object + host program
set -e
cd "$(dirname "$0")"
hipcc --cuda-device-only --no-gpu-bundle-output --offload-arch=gfx950 -O3 -mcode-object-version=4 -c kernel.hip -o kernel.hsaco
hipcc -O2 probe.cpp -o probe -L/opt/rocm/lib -lhsa-runtime64
