"""Front-end disk cache in the reference's on-disk format."""
