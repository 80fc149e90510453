"""Builder surface of the reference's `ops/` package (layers, activations, input)."""
