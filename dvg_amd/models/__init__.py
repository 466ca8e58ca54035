"""Drop-in counterparts of the reference's `models/` package (same module, class and
constructor names) running on the MI355X kernel library."""
