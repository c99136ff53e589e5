#!/bin/bash
# ISA gate (see tools/check_isa.py): builds if needed, then fails on VGPR spills / scratch outside the allow-list.
cd "$(dirname "$0")/.." && bash simxns_amd/csrc/build.sh >/dev/null && exec python tools/check_isa.py "$@"
