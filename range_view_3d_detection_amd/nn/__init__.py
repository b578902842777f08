"""Neural-network modules (mirror of ``torchbox3d.nn``)."""
