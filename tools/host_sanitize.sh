#!/bin/bash
# AddressSanitizer + UBSan, then ThreadSanitizer, over the host-only sources (no GPU involved)
set -e
cd "$(dirname "$0")/.."
out=$(mktemp -d)
src="tools/host_sanitize.cpp hpfw_amd/csrc/plan.cpp hpfw_amd/csrc/eigen_host.cpp"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Ihpfw_amd/csrc -Iinclude $src -o $out/asan -lpthread
$out/asan
g++ -std=c++17 -O1 -g -fsanitize=thread -Ihpfw_amd/csrc -Iinclude $src -o $out/tsan -lpthread
$out/tsan
rm -rf $out
echo "sanitizers: clean"
