"""CPU parity oracle (test infrastructure).  Importable only from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never from the product package."""
