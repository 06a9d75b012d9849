"""sfm_amd: MI355X-native matching + bundle-adjustment hot path."""
