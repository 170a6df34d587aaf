class SeqRecord(object):
    def __init__(self, seq, id="", description="", name=""):
        self.seq, self.id, self.description, self.name = seq, id, description, name

    def __len__(self):          # Biopython: the length of the sequence
        return len(self.seq)
