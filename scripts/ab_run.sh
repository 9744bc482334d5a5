#!/bin/bash
# A/B on the GPU box: run "$@" with the normal library, then with libfi_hip_exp.so in its place.
cd "$(dirname "$0")/.."
L=feature_intertwiner_amd/libfi_hip.so
cp $L /tmp/fi_base.so
echo "=== base"; "$@"
cp feature_intertwiner_amd/libfi_hip_exp.so $L
echo "=== exp"; "$@"
cp /tmp/fi_base.so $L
echo "=== base again"; "$@"
