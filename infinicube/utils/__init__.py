# drop-in namespace shim for the utilities this repo provides MI355X-native versions of
