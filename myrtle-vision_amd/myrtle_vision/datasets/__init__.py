"""CPU data layer (PIL + numpy; the reference used torchvision, which is not a dependency here)."""
