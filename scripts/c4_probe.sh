#!/bin/bash
python scripts/c4_probe.py
for d in $@; do
  FI_DBG_1X1=$d python scripts/c4_probe.py
done
