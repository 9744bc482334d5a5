for e in 1 4 129; do
  rm -f feature_intertwiner_amd/csrc/nms.o; FI_EXTRA_HIPCC_FLAGS=-DFI_NMS_EXP=$e python -m feature_intertwiner_amd.build > /dev/null 2>&1
  echo "EXP $e"; python scripts/nms_scan_probe.py 2>&1 | grep '"batch": 1' | head -n 1
done
rm -f feature_intertwiner_amd/csrc/nms.o; python -m feature_intertwiner_amd.build > /dev/null 2>&1
