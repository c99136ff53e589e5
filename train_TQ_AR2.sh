#!/bin/bash
# TQ recipe of the reference (SimANS/train_TQ_AR2.sh): loop and hyper-parameters in simxns_amd/launch.py
exec python -u -m simxns_amd.launch TQ "$@"
