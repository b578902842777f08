"""Math ops (mirror of ``torchbox3d.math``) backed by the HIP library."""
