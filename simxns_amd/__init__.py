"""simxns_amd -- MI355X-native engine for the SimANS/co_training bi-encoder hot path."""
__version__ = "0.1.0"
