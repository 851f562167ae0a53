"""constants the real pic1dp_amd/parallel.py takes from the binding (stand-in package: tests/fake_engine)"""
COMM_ID_BYTES = 128
XCHG_HANDLE_BYTES = 64
