"""Mirror of the reference's ``uda`` package (model zoo only; the CPU dataset layer is out of scope)."""
