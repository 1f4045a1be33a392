#!/bin/bash
export PIRGPU_ALLOW_ENV=1   # the library reads PIRGPU_* knobs only behind this gate (csrc/env_gate.h)
# Fuzz the request parser (wire.cpp + wire_codec.cpp, device-free validation entry) under ASan + UBSan.
# Host only; run from the repo root:  bash tools/fuzz_wire_asan.sh
set -e
g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    pir_amd/csrc/wire.cpp pir_amd/csrc/wire_codec.cpp tools/wire_stubs.cpp -o /tmp/libwire_asan.so
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0 \
    WIRE_ASAN_LIB=/tmp/libwire_asan.so python3 tools/fuzz_wire_asan.py
