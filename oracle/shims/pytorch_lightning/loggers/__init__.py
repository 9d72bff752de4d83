class CSVLogger:
    def __init__(self, *a, **k):
        pass
