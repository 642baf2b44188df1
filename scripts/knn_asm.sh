#!/bin/bash
# Device assembly of the screened k-NN kernel (d <= 16, r <= 16 instance) with its register / scratch / LDS summary.
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o /tmp/asm/knn.s flgp_amd/csrc/knn.hip 2>&1 | grep -E "error" -A5 || true
awk '/^_ZN4flgp17knn_screen_kernelILi16ELi16E/{f=1} f{print} f&&/^\.Lfunc_end/{exit}' /tmp/asm/knn.s > /tmp/asm/scr.s
awk '/^_ZN4flgp17knn_screen_kernelILi16ELi16E/{f=1} f&&/NumVgprs|Occupancy|ScratchSize|NumAgprs|LDSByte/{print} f&&/Occupancy/{exit}' /tmp/asm/knn.s
