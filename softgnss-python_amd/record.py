"""IF records resident in HBM, presented with the reference's two input shapes.

The reference hands acquisition an int8 ndarray (initialize.py:481) and tracking an open file
(tracking.py:107,154,255).  These wrappers let a record that already lives on the GPU (uploaded
once, or generated there) be passed to the same methods without a round trip through the host.
"""


class DeviceSignal(object):
    """`longSignal` for AcquisitionResult.acquire: samples [offset, offset+length) of a record."""

    def __init__(self, record, offset=0, length=None):
        self.record = record
        self.offset = int(offset)
        self.length = int(len(record) - offset if length is None else length)
        if self.offset < 0 or self.length < 0 or self.offset + self.length > len(record):
            raise ValueError("window outside the record")

    def __len__(self):
        return self.length


class DeviceFile(object):
    """`fid` for TrackingResult.track: file-like view (seek/tell/close) of a device record.

    file_offset is the byte offset, in the file the reference would read, of the record's first
    sample, so positions reported by tell() / absoluteSample are file positions.
    """

    def __init__(self, record, file_offset=0):
        self.record = record
        self.file_offset = int(file_offset)
        self._pos = self.file_offset
        self.closed = False

    def seek(self, off, whence=0):
        if whence != 0:
            raise ValueError("only absolute seeks")
        self._pos = int(off)       # tracking.py:107 passes a float64 offset
        return self._pos

    def tell(self):
        return self._pos

    def close(self):
        self.closed = True
