#!/bin/bash
# MS_Pas recipe of the reference (SimANS/train_MS_Pas_AR2.sh): loop and hyper-parameters in simxns_amd/launch.py
exec python -u -m simxns_amd.launch MS_Pas "$@"
