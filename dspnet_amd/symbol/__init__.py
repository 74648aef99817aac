"""Graph definitions of the DSPNet hot path, named after the reference's symbol/ package."""
