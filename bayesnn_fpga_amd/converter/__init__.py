"""Mirror of the reference's ``Hardware_Artifact/converter`` for the PyTorch side."""
