"""CPU oracle for the sample-batch path - TEST INFRASTRUCTURE, not product (see oracle/README.md)."""
