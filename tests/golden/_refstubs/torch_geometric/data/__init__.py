class Data:  # utils/field.py:6 imports the name only
    pass
