"""Mirror of the reference's ``enf`` package layout for the accelerated path."""
