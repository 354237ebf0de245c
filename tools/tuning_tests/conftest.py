def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X and a TUNING=1 build of libvsgpu.so")
