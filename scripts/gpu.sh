#!/bin/bash
# build the library here, then run a command on an MI355X box:  scripts/gpu.sh [timeout_s] '<command>'
set -e
cd "$(dirname "$0")/.."
make -C eyoc_amd/csrc -j8 2>&1 | grep -E "error|warning: unused|Error" || true
test -f eyoc_amd/lib/libeyoc_hip.so
T=${1:-900}; shift || true
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
