"""Drop-in counterparts of catch.filter.* for the accelerated path."""
