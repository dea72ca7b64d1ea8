"""Host side of the MI355X-native ADNM-UNet hot path: ctypes binding of libadnm_hip.so
(`lib`), autograd wrappers over its kernels (`ops`), the deterministic parameter / input
recipe (`recipe`) and the data-parallel gradient bucket (`ddp`)."""
