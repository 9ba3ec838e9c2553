"""nerfool_amd: MI355X-native adversarial inner loop of NeRFool."""
