"""MI355X-native hot path of the ADMM / FTRL sparse logistic-regression trainer.

capi   -- ctypes binding of libadmm_hip.so (the C ABI of include/admm_hip.h)
driver -- host-side mirror of the reference's Worker / Master / ArgParser on top of it
build  -- hipcc build recipe
"""
