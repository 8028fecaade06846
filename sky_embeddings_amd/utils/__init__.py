"""Host-side mirror of the reference's ``utils`` package for the hot path (same names,
argument meaning and error behaviour)."""
