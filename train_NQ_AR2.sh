#!/bin/bash
# NQ recipe of the reference (SimANS/train_NQ_AR2.sh): loop and hyper-parameters in simxns_amd/launch.py
exec python -u -m simxns_amd.launch NQ "$@"
