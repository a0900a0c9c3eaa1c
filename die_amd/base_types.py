"""Channel names of the data model (reference core/base_types.py:31-36)."""


class DataChannels:
    medium = ('agents', 'env_food', 'chem1')
    agents = ('x', 'y', 'alive', 'agent_food')
    actions = ('dx', 'dy', 'deposit1')
