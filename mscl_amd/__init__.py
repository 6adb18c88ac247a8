"""mscl_amd -- MI355X-native MSCL training hot path."""
__version__ = '0.1.0'
