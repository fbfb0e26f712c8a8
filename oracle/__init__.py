"""CPU oracles of the NeRF-SR paths (test infrastructure only: never imported by the product)."""
