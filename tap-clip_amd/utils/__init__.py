"""Host-side helpers mirroring the reference's `utils/` package."""
