"""Training step of the denoising path on MI355X: autograd Functions whose forward AND backward are HIP
kernel launches (functions.py), the UNet walked with them (unet.py), the v-prediction loss, AdamW and the data-parallel
gradient all-reduce (step.py).  Reference: lvdm/models/ddpm3d.py:741-802,1267-1300; main/utils_train.py:126-137."""
