"""Host-side mirror of the reference's ``models`` package (same class and FACTORY names)."""
